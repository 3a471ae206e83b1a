#!/bin/bash
# same-box A/B of the per-shape non-temporal stores in conv_igemm's wide 3x3 epilogue (SRGANFD_CONV_NT=1 default, 0 = plain stores), alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in 0 1; do
    SRGANFD_CONV_NT=$v python bench.py --workload both --no-cpu-baseline --no-module-loop --no-kernel-events 2> gpurun_out/conv_nt_$v.err | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('CONV_NT=$v g_only', r['ms_per_step'], 'gan', r['gan']['ms_per_step'])" || { tail -5 gpurun_out/conv_nt_$v.err; exit 1; }
  done
done
