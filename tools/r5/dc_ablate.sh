#!/bin/bash
for d in 15 7; do echo "SRGANFD_DC_DBG=$d"; SRGANFD_DC_DBG=$d python tools/r5/dense_chain_bench.py 1 8 32 1 32 32 4 32 32 16 32 32 2>&1 | grep -v amdgpu.ids | grep fwd; done
