#!/bin/bash
# EXPERIMENT (needs a temporary SRGANFD_WGRAD_SPLITS switch in ops.WgradPlan: `if splits == 0 and x_channels == 192: splits = int(os.environ[...])`): pixel splits of the dense-block weight gradient inside the small-shape steps; result in profiles/r05_wgrad_splits_small_shapes.txt: 36 (one workgroup per CU) stays
out=gpurun_out/r5_e2e_wgsplits.txt; : > $out
for rep in 1 2; do
for w in "--workload g_only --batch 16 --lr-size 32" "--workload g_only --batch 4 --lr-size 32" "--workload aesrgan_gan --batch 8 --lr-size 60 --upscale 2"; do
  for sp in 36 24 18 12; do
    x=$(SRGANFD_WGRAD_SPLITS=$sp timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" 2>&1)
    echo "$w  SRGANFD_WGRAD_SPLITS=$sp  ms/step img/s: $x" | tee -a $out
  done
done
done
