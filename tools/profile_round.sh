#!/bin/bash
# Round profiles of bench.py's own command (run on the GPU box): per-kernel time (rocprofv3 --kernel-trace --stats) and two PMC passes
# (fabric-side reads / writes; MFMA-busy), each in its own run as the micro-architecture guide prescribes.
#   tools/profile_round.sh <out dir under gpurun_out> <workload> [extra bench args]
out=$GRAFT_REPO_ROOT/$1; wl=$2; shift 2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python3 $B > $out/stats_$wl.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $out/pmc_tcc_$wl -- python3 $B > $out/pmc_tcc_$wl.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/pmc_sq_$wl -- python3 $B > $out/pmc_sq_$wl.log 2>&1
ls $out
