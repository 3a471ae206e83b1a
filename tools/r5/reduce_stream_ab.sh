#!/bin/bash
# same-box alternating A/B of the slab reductions on a side stream (SRGANFD_REDUCE_STREAM=1) at the reference-default shapes and the headline shape
out=gpurun_out/r5_reduce_stream_ab.txt; : > $out
for rep in 1 2; do
for w in "--workload g_only --batch 16 --lr-size 32" "--workload g_only --batch 16 --lr-size 48" "--workload g_only --batch 16 --lr-size 72" "--workload esrgan_gan --batch 16" "--workload gan --batch 16 --lr-size 72 --upscale 2" "--workload g_only" "--workload gan"; do
  for v in 0 1; do
    r=$(SRGANFD_REDUCE_STREAM=$v timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('dense_chain'))" 2>&1) || exit 1
    echo "$w  SRGANFD_REDUCE_STREAM=$v  ms/step img/s: $r" | tee -a $out
  done
done
done
