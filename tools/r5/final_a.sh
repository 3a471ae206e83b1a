#!/bin/bash
# round-5 evidence, part A: the GPU suite, the dense-chain end-to-end A/B, the reference-default shapes with the final tree
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r05_gpu_suite.txt; cat gpurun_out/r05_gpu_suite.txt
bash tools/r5/e2e_chain_ab.sh > /dev/null 2>&1; cp gpurun_out/r5_e2e_chain_ab.txt gpurun_out/r05_dense_chain_e2e_ab.txt; cat gpurun_out/r05_dense_chain_e2e_ab.txt
bash tools/r5_shapes.sh gpurun_out/r05_reference_shapes_after.txt > /dev/null 2>&1; grep "^==\|ms_per_step" gpurun_out/r05_reference_shapes_after.txt | cut -c1-200
