#!/bin/bash
# EXPERIMENT: cache policy of conv_igemm's epilogue stores (variant libraries: st1 = sc1 write-through, st2 = sc0 sc1) against the product, alternating.
# Idea: a kernel that leaves up to 32 MB of dirty L2 lines pays their write-back at its end; write-through stores would spread it over the kernel.
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for v in hip st1 st2; do
    SRGANFD_LIB=$GRAFT_REPO_ROOT/sr_gan_fd_amd/libsrganfd_$v.so python bench.py --workload both --no-cpu-baseline --no-module-loop --no-kernel-events 2> gpurun_out/st_$v.err | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v g_only', r['ms_per_step'], 'gan', r['gan']['ms_per_step'])" || { tail -5 gpurun_out/st_$v.err; exit 1; }
  done
done
