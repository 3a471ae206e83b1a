#!/bin/bash
# same-box alternating A/B of the in-tree library ("new") against gpurun_in/libold.so ("old") on training steps: tools/r5/lib_ab.sh <out> <rounds> "<bench args>" ...
out=$1; rounds=$2; shift 2; : > $out
R=$GRAFT_REPO_ROOT
for rep in $(seq $rounds); do
for w in "$@"; do
  for v in old new; do
    if [ $v = old ]; then export SRGANFD_LIB=$R/gpurun_in/libold.so; else unset SRGANFD_LIB; fi
    r=$(timeout -k 10 300 python bench.py $w --steps 30 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_classes',{}).get('wgrad_reduce_batch',{}); print(d['ms_per_step'], d['value'], 'reduce_batch avg us', k.get('avg_us'))" 2>&1) || exit 1
    echo "$w  $v  ms/step img/s: $r" | tee -a $out
  done
done
done
