#!/bin/bash
# same-box alternating A/B of the 8-row tiles of the 32-channel conv (srganfd_set_conv_rows8 / SRGANFD_CONV_ROWS8) at the shapes whose launches
# are one partial round of 16-row tiles: batch 16 at 72 x 72 (bsrnet_config.py:69-70, bsrgan_config.py:101-102), and two controls
out=gpurun_out/r5_rows8_ab.txt; : > $out
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
for k,v in sorted(d.get('kernel_classes',{}).items(), key=lambda kv:-kv[1]['ms'])[:$1]:
    print('      %6.2f ms %6d launches %8.1f us  %s' % (v['ms'], v['launches'], v['avg_us'], k))
"; }
for rep in 1 2; do
for w in "--workload g_only --batch 16 --lr-size 72" "--workload gan --batch 16 --lr-size 72 --upscale 2" "--workload g_only --batch 8 --lr-size 96"; do
  for r8 in 0 auto; do
    echo "$w  CONV_ROWS8=$r8  ms/step img/s:" >> $out
    SRGANFD_CONV_ROWS8=$r8 timeout -k 10 300 python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop --no-bf16 --no-kernel-events 2>/dev/null | line 0 >> $out || exit 1
  done
done
done
# per-kernel view (HIP events per launch), once each
for r8 in 0 auto; do
  echo "== layer classes, g_only batch 16 72 -> 288, CONV_ROWS8=$r8" >> $out
  SRGANFD_CONV_ROWS8=$r8 timeout -k 10 300 python bench.py --workload g_only --batch 16 --lr-size 72 --steps 10 --warmup 3 --no-cpu-baseline --no-module-loop --no-bf16 2>/dev/null | line 6 >> $out || exit 1
done
cat $out
