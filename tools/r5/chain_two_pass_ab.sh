#!/bin/bash
# same-box alternating A/B at batch 16, 72 x 72 (bsrnet_config.py:69-70, bsrgan_config.py:101-102): per-layer launches / dense-block launch in two passes of
# 16 x 16 tiles (250 + 150) / of 12 x 16 tiles (240 + 240) / per-layer launches with the weight gradients on a side stream
out=gpurun_out/r5_chain_two_pass_ab.txt; : > $out
for rep in 1 2; do
for w in "--workload g_only --batch 16 --lr-size 72" "--workload gan --batch 16 --lr-size 72 --upscale 2"; do
  for v in "SRGANFD_DENSE_CHAIN=0" "SRGANFD_DENSE_CHAIN=1" "SRGANFD_DENSE_CHAIN=1 SRGANFD_DC_RPW=3" "SRGANFD_DENSE_CHAIN=0 SRGANFD_WGRAD_STREAM=1"; do
    r=$(env $v timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('dense_chain'))" 2>&1) || exit 1
    echo "$w  $v  ms/step img/s: $r" | tee -a $out
  done
done
done
