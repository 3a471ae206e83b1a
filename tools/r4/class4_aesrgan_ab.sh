#!/bin/bash
# same-box A/B of the one-launch stride-2 data gradients in the A-ESRGAN step (3x3 stride-2 encoder, 2x2 stride-2 attention gates)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for v in 0 1; do
    SRGANFD_CLASS4=$v python bench.py --workload aesrgan_gan --no-cpu-baseline --no-kernel-events 2> gpurun_out/class4a_$v.err | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('CLASS4=$v aesrgan', r['ms_per_step'], r['value'])" || { tail -5 gpurun_out/class4a_$v.err; exit 1; }
  done
done
