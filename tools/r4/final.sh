#!/bin/bash
# round-4 evidence: kernel stats + PMC passes of both default workloads, the other workloads' lines, the default bench line
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh gpurun_out/prof_r04 g_only > /dev/null 2>&1 || exit 1
echo "g_only profiled"
bash $R/tools/profile_round.sh gpurun_out/prof_r04 gan > /dev/null 2>&1 || exit 1
echo "gan profiled"
cd $R
for wl in aesrgan_gan esrgan_gan; do
  python bench.py --workload $wl --no-cpu-baseline > gpurun_out/r04_${wl}_b32_bench.json 2> gpurun_out/r04_${wl}.err || { tail -5 gpurun_out/r04_${wl}.err; exit 1; }
  echo "$wl done"
done
python bench.py --workload realesrgan_gan --batch 48 --no-cpu-baseline > gpurun_out/r04_realesrgan_gan_b48_bench.json 2> gpurun_out/r04_realesrgan.err || { tail -5 gpurun_out/r04_realesrgan.err; exit 1; }
python tools/layer_table.py --workload gan > gpurun_out/r04_layer_table_gan.txt 2>/dev/null
python tools/layer_table.py --workload g_only > gpurun_out/r04_layer_table_g_only.txt 2>/dev/null
python bench.py > gpurun_out/r04_default_bench.json 2> gpurun_out/r04_default_bench.err || { tail -5 gpurun_out/r04_default_bench.err; exit 1; }
tail -c 600 gpurun_out/r04_default_bench.json
