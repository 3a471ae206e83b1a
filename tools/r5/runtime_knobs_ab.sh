#!/bin/bash
# zero-code probe of HIP runtime dispatch knobs at a launch-latency-bound shape (batch 16, 72 x 72) and the headline shape; alternating, two rounds
out=gpurun_out/r5_runtime_knobs.txt; : > $out
for rep in 1 2; do
for w in "--workload g_only --batch 16 --lr-size 72" "--workload g_only --batch 16 --lr-size 32" "--workload g_only"; do
  for v in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "DEBUG_HIP_KERNARG_COPY_OPT=1" "AMD_DIRECT_DISPATCH=0"; do
    r=$(env $v timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" 2>&1) || r="failed"
    echo "$w  $v  ms/step img/s: $r" | tee -a $out
  done
done
done
