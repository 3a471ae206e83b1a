#!/bin/bash
# conv_stream (LDS-DMA streaming conv, one persistent workgroup per CU) against conv_igemm at the reference-default shapes
#   tools/r5/stream_ab.sh <out file>
out=$1; : > $out
run() { echo "== SRGANFD_USE_STREAM=$S $*" >> $out; SRGANFD_USE_STREAM=$S python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.3f ms/step  %.1f img/s' % (d['ms_per_step'], d['value']))
for k,v in sorted(d.get('kernel_classes',{}).items(), key=lambda kv:-kv[1]['ms'])[:6]:
    print('   %6.2f ms %6d launches %8.1f us  %s' % (v['ms'], v['launches'], v['avg_us'], k))
" >> $out; }
for S in 0 1; do
run --workload g_only --batch 16 --lr-size 72
run --workload g_only --batch 16 --lr-size 48
run --workload esrgan_gan --batch 16
run --workload gan --batch 16 --lr-size 72 --upscale 2
done
S=1 run --workload g_only
cat $out
