#!/bin/bash
# EXPERIMENT: dense-block weight gradients on a side stream (SRGANFD_WGRAD_STREAM=1, engine.py) beside the dense-chain data-gradient launches at the small shapes,
# where the chain occupies at most half of the CUs
out=gpurun_out/r5_e2e_wgstream.txt; : > $out
for rep in 1 2; do
for w in "--workload g_only --batch 16 --lr-size 32" "--workload g_only --batch 16 --lr-size 48" "--workload esrgan_gan --batch 16" "--workload g_only --batch 4 --lr-size 32"; do
  for ws in 0 1; do
    x=$(SRGANFD_WGRAD_STREAM=$ws timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('dense_chain'))" 2>&1)
    echo "$w  SRGANFD_WGRAD_STREAM=$ws  ms/step img/s: $x" | tee -a $out
  done
done
done
