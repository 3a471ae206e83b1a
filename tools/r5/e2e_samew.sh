#!/bin/bash
# EXPERIMENT (timing only, results wrong): every dense block reads block 0's packed weights (needs a temporary switch in engine.py: `_SW = (lambda i: 0) if os.environ.get("SRGANFD_SAME_W") == "1" else (lambda i: i)` around the block index of the four `pk["offs"][("f" | "b", i, ...)]` lookups of the dense blocks; result: profiles/r05_weight_prefetch_upper_bound.txt), so the
# dense-chain launches find their 0.96 MB weight stream in L2: the upper bound of what prefetching the next block's weights could give
out=gpurun_out/r5_e2e_samew.txt; : > $out
for round in 1 2; do
for w in "--workload g_only --batch 16 --lr-size 32" "--workload g_only --batch 4 --lr-size 32" "--workload g_only --batch 16 --lr-size 48"; do
  for sw in 0 1; do
    x=$(SRGANFD_SAME_W=$sw timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" 2>&1)
    echo "$w  SRGANFD_SAME_W=$sw  ms/step img/s: $x" | tee -a $out
  done
done
done
