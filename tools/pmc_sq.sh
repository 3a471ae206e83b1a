#!/bin/bash
# SQ / LDS / TA counter passes over tools/kbench.py for one kernel choice: tools/pmc_sq.sh <mode> <outdir> [kbench args]
mode=$1; out=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TD_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/sq_m${mode}_p$i -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --modes $mode --rounds 1 --reps 3 "$@" > $GRAFT_REPO_ROOT/$out/sq_m${mode}_p$i.log 2>&1
done
