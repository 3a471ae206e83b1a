#!/bin/bash
# round-3 session 1: box baseline, workgroup-stagger experiment on the dense-block convs, L2 hit rate of the weight-gradient launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp1; mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_g_only.json 2> $O/bench_g_only.err || exit 1
tail -c 600 $O/bench_g_only.json
SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so python tools/kbench.py --dtype f16 --modes 8 --dbg 0,384,640,1152,2176 --rounds 4 --reps 20 > $O/kbench_stagger.txt 2>&1 || { tail -5 $O/kbench_stagger.txt; exit 1; }
cat $O/kbench_stagger.txt
python tools/wgbench.py --dtype f16 --variants 3 --splits 0,16,64 > $O/wgbench.txt 2>&1 || { tail -5 $O/wgbench.txt; exit 1; }
cat $O/wgbench.txt
cd /tmp && export TMPDIR=/tmp
for ctr in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $ctr | tr ' ' '_')
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_wg_$tag -- python3 $R/tools/wgbench.py --dtype f16 --variants 3 --splits 0 > $O/pmc_wg_$tag.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $O/pmc_wg_$tag | head -20
done
