#!/bin/bash
# per-kernel time of the sibling workloads (rocprofv3 --kernel-trace --stats of bench.py's own command)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for spec in "aesrgan_gan 32" "esrgan_gan 32" "realesrgan_gan 48"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_other/$1 -- python3 $R/bench.py --workload $1 --batch $2 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/prof_other/$1.log 2>&1 || { tail -5 $R/gpurun_out/prof_other/$1.log; exit 1; }
  echo "$1 done"
done
