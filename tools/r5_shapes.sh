#!/bin/bash
# Reference-default shapes: step time of the fused trainers at the batch / crop sizes the reference's configs ship.
#   bsrgan_config.py:62,101-102  x2, LR 72 -> 144, batch 16 (GAN)      bsrnet_config.py:55,69-70  x4, 72 -> 288, batch 16 (generator only)
#   esrgan_config.py:73-74       x4, 32 -> 128, batch 16 (GAN)         rrdbnet_config.py:51-52    x4, 48 -> 192, batch 16 (generator only)
#   aesrgan_config.py:62,102-103 x2, 60 -> 120, batch 8 (GAN)
# usage: tools/r5_shapes.sh <out file under gpurun_out>
out=$1
: > $out
run() { echo "== $*" >> $out; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-module-loop "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('metric','value','ms_per_step','step_tflops_per_gpu','config')}))
r=d.get('roofline',{})
print('  dominant:', r.get('kernel'), 'avg_us', r.get('avg_launch_us'), 'mfma_frac', r.get('mfma_frac'), 'hbm_frac', r.get('hbm_frac'))
for k,v in sorted(d.get('kernel_classes',{}).items(), key=lambda kv:-kv[1]['ms'])[:8]:
    print('   %6.2f ms %6d launches %8.1f us  %s' % (v['ms'], v['launches'], v['avg_us'], k))
" >> $out; }
run --workload gan --batch 16 --lr-size 72 --upscale 2
run --workload g_only --batch 16 --lr-size 72
run --workload esrgan_gan --batch 16
run --workload g_only --batch 16 --lr-size 48
run --workload aesrgan_gan --batch 8 --lr-size 60 --upscale 2
run --workload gan --batch 16 --lr-size 72
cat $out
