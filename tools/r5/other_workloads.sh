#!/bin/bash
# round-5 lines of the other workloads (final tree): A-ESRGAN GAN at configs[4]'s per-GPU shard, ESRGAN relativistic iteration, Real-ESRGAN iteration
cd $GRAFT_REPO_ROOT
for wl in aesrgan_gan esrgan_gan; do
  python bench.py --workload $wl --no-cpu-baseline --no-module-loop --no-bf16 > gpurun_out/r05_${wl}_b32_bench.json 2> gpurun_out/r05_${wl}.err || { tail -5 gpurun_out/r05_${wl}.err; exit 1; }
  python -c "import json; d=json.loads(open('gpurun_out/r05_${wl}_b32_bench.json').read().strip().splitlines()[-1]); print('$wl', d['ms_per_step'], d['value'])"
done
python bench.py --workload realesrgan_gan --batch 48 --no-cpu-baseline --no-module-loop --no-bf16 > gpurun_out/r05_realesrgan_gan_b48_bench.json 2> gpurun_out/r05_realesrgan.err || { tail -5 gpurun_out/r05_realesrgan.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/r05_realesrgan_gan_b48_bench.json').read().strip().splitlines()[-1]); print('realesrgan_gan b48', d['ms_per_step'], d['value'])"
