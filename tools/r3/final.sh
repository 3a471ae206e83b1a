#!/bin/bash
# round-3 final evidence: kernel stats + PMC passes of both default workloads, then the default bench line
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh gpurun_out/prof_r03 g_only > /dev/null 2>&1 || exit 1
echo "g_only profiled"
bash $R/tools/profile_round.sh gpurun_out/prof_r03 gan > /dev/null 2>&1 || exit 1
echo "gan profiled"
cd $R && python bench.py > gpurun_out/r03_default_bench.json 2> gpurun_out/r03_default_bench.err || { tail -5 gpurun_out/r03_default_bench.err; exit 1; }
tail -c 1500 gpurun_out/r03_default_bench.json
