#!/bin/bash
# same-box A/B of the one-launch stride-2 data gradient (SRGANFD_CLASS4=1, default) against four class launches (=0), alternating
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for v in 0 1; do
    SRGANFD_CLASS4=$v python bench.py --workload gan --no-cpu-baseline --no-module-loop 2> gpurun_out/class4_$v.err | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('CLASS4=$v', r['ms_per_step'], r['value'])" || { tail -5 gpurun_out/class4_$v.err; exit 1; }
  done
done
python tools/layer_table.py --workload gan 2>/dev/null | grep -i "k2 s1\|classes" 
