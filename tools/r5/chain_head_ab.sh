#!/bin/bash
# same-box alternating A/B of a dense_chain.hip change: in-tree library ("new") against gpurun_in/libold.so ("old"): the launch alone, then the steps that take it
out=gpurun_out/r5_chain_head_ab.txt; : > $out
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export SRGANFD_LIB=$R/gpurun_in/libold.so; else unset SRGANFD_LIB; fi
    echo "== $v: dense_chain_bench" >> $out
    timeout -k 10 200 python tools/r5/dense_chain_bench.py 16 32 32 16 48 48 8 60 60 4 32 32 2>/dev/null >> $out || exit 1
  done
done
for rep in 1 2 3; do
for w in "--workload g_only --batch 16 --lr-size 32" "--workload g_only --batch 4 --lr-size 32" "--workload g_only --batch 16 --lr-size 48" "--workload esrgan_gan --batch 16"; do
  for v in old new; do
    if [ $v = old ]; then export SRGANFD_LIB=$R/gpurun_in/libold.so; else unset SRGANFD_LIB; fi
    r=$(timeout -k 10 300 python bench.py $w --steps 30 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('dense_chain'))" 2>&1) || exit 1
    echo "$w  $v  ms/step img/s: $r" >> $out
  done
done
done
cat $out
