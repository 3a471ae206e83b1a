#!/bin/bash
# EXPERIMENT (profiles/r05_plan.txt item e): dense-block weight gradients on a CU-MASKED side stream (half of every XCD) beside the dense-chain
# data-gradient launches; same box, alternating, two rounds.  The whole run on a non-blocking stream (SRGANFD_BENCH_OWN_STREAM=1): a CU-masked stream is a
# blocking stream and serialises with the null stream (first attempt: every masked variant 2x slower, profiles/r05_cu_partitioned_backward.txt).  Result: profiles/r05_cu_partitioned_backward.txt
out=gpurun_out/r5_e2e_cumask.txt; : > $out
HALF=ffffffff,ffffffff,ffffffff,ffffffff,0,0,0,0
QUARTER=ffffffff,ffffffff,0,0,0,0,0,0
run() {  # workload-args, label, env...
  local w="$1" label="$2"; shift 2
  x=$(env SRGANFD_BENCH_OWN_STREAM=1 "$@" timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('dense_chain'))" 2>&1)
  echo "$w  [$label]  ms/step img/s: $x" | tee -a $out
}
for round in 1 2; do
for w in "--workload g_only --batch 16 --lr-size 32" "--workload g_only --batch 4 --lr-size 32" "--workload esrgan_gan --batch 16" "--workload aesrgan_gan --batch 8 --lr-size 60 --upscale 2"; do
  run "$w" "main stream only (null stream)" SRGANFD_WGRAD_STREAM=0 SRGANFD_BENCH_OWN_STREAM=0
  run "$w" "main stream only" SRGANFD_WGRAD_STREAM=0
  run "$w" "side stream, no mask" SRGANFD_WGRAD_STREAM=1
  run "$w" "weight gradient on 128 CUs (18 splits), reductions on a third stream, 4 blocks per reduction" SRGANFD_WGRAD_STREAM=1 SRGANFD_WGRAD_CUMASK=$HALF SRGANFD_WGRAD_SPLITS=18
  run "$w" "... hand-over every 3 blocks" SRGANFD_WGRAD_STREAM=1 SRGANFD_WGRAD_CUMASK=$HALF SRGANFD_WGRAD_SPLITS=18 SRGANFD_WGRAD_GROUP=3
  run "$w" "... hand-over every 6 blocks" SRGANFD_WGRAD_STREAM=1 SRGANFD_WGRAD_CUMASK=$HALF SRGANFD_WGRAD_SPLITS=18 SRGANFD_WGRAD_GROUP=6
  run "$w" "side stream, no mask, hand-over every 3 blocks" SRGANFD_WGRAD_STREAM=1 SRGANFD_WGRAD_GROUP=3
done
done
for v in "SRGANFD_WGRAD_STREAM=0" "SRGANFD_WGRAD_STREAM=1 SRGANFD_WGRAD_CUMASK=$HALF SRGANFD_WGRAD_SPLITS=18 SRGANFD_WGRAD_GROUP=3"; do
  env SRGANFD_BENCH_OWN_STREAM=1 $v timeout -k 10 200 python tools/r5/host_time.py 2>&1 | tail -1 | tee -a $out
done
