#!/bin/bash
# EXPERIMENT: non-temporal LOADS of the epilogue operands (residual / mask) in the NT twin of the wide 3x3 kinds: variant library against the product, alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in prod ntld; do
    L=$GRAFT_REPO_ROOT/sr_gan_fd_amd/libsrganfd_hip.so; [ $v = ntld ] && L=$GRAFT_REPO_ROOT/sr_gan_fd_amd/libsrganfd_ntld.so
    SRGANFD_LIB=$L python bench.py --workload gan --no-cpu-baseline --no-module-loop --no-kernel-events 2> gpurun_out/ntld_$v.err | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v gan', r['ms_per_step'])" || { tail -5 gpurun_out/ntld_$v.err; exit 1; }
  done
done
