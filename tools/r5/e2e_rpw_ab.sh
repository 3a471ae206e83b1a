#!/bin/bash
# end-to-end: the dense-chain tile geometry (rows per wave 4 / 2 / 1) at the reference's default shapes, same box, alternating
out=gpurun_out/r5_e2e_rpw_ab.txt; : > $out
for rep in 1 2; do
for w in "--workload g_only --batch 16 --lr-size 48" "--workload g_only --batch 16 --lr-size 32"; do
  for r in 4 3 2; do
    x=$(SRGANFD_DC_RPW=$r timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" 2>&1)
    echo "$w  SRGANFD_DC_RPW=$r  ms/step img/s: $x" | tee -a $out
  done
done
done
