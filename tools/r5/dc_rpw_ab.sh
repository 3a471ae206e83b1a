#!/bin/bash
# dense-chain launch: 8 x 16 tiles (SRGANFD_DC_RPW=2, the automatic choice when the batch is one pass that way) against 16 x 16 (=4)
timeout -k 10 420 python -m pytest tests/test_dense_chain_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for r in 2 3; do SRGANFD_DC_RPW=$r timeout -k 10 420 python -m pytest tests/test_dense_chain_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3; done
for r in 4 3 2; do echo "SRGANFD_DC_RPW=$r"; SRGANFD_DC_RPW=$r timeout -k 10 200 python tools/r5/dense_chain_bench.py 16 32 32 8 60 60 4 32 32 16 48 48 2>&1 | grep -v amdgpu.ids; done
