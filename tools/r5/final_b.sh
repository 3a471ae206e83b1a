#!/bin/bash
# round-5 evidence, part B: kernel stats + PMC passes of both default workloads, layer tables, the default bench line
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh gpurun_out/prof_r05 g_only > /dev/null 2>&1 || exit 1
echo "g_only profiled"
bash $R/tools/profile_round.sh gpurun_out/prof_r05 gan > /dev/null 2>&1 || exit 1
echo "gan profiled"
cd $R
python tools/layer_table.py --workload gan > gpurun_out/r05_layer_table_gan.txt 2>/dev/null
python tools/layer_table.py --workload g_only > gpurun_out/r05_layer_table_g_only.txt 2>/dev/null
python bench.py > gpurun_out/r05_default_bench.json 2> gpurun_out/r05_default_bench.err || { tail -5 gpurun_out/r05_default_bench.err; exit 1; }
tail -c 1500 gpurun_out/r05_default_bench.json
