#!/bin/bash
# end-to-end A/B of the LDS-resident dense-block launch at the reference's default shapes: the same bench line with SRGANFD_DENSE_CHAIN=0 / auto, alternating
out=gpurun_out/r5_e2e_chain_ab.txt; : > $out
for rep in 1 2; do
for w in "--workload esrgan_gan --batch 16" "--workload g_only --batch 16 --lr-size 32" "--workload g_only --batch 16 --lr-size 48" "--workload aesrgan_gan --batch 8 --lr-size 60 --upscale 2"; do
  for dc in 0 auto; do
    r=$(SRGANFD_DENSE_CHAIN=$dc timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" 2>&1)
    echo "$w  DENSE_CHAIN=$dc  ms/step img/s: $r" | tee -a $out
  done
done
done
