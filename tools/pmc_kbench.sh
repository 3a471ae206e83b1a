#!/bin/bash
# PMC passes over tools/kbench.py (one mode per pass): fabric-side reads / writes and L2 hit rate per kernel.
# usage: tools/pmc_kbench.sh <mode> <outdir> [kbench args]
mode=$1; out=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for ctr in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctr | tr ' ' '_')
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_m${mode}_$tag -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --modes $mode --rounds 1 --reps 3 "$@" > $GRAFT_REPO_ROOT/$out/pmc_m${mode}_$tag.log 2>&1
done
