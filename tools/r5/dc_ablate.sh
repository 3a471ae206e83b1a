#!/bin/bash
# EXPERIMENT: epilogue ablations of the dense-chain launch (variant library built from tools/experiments/r5_dense_chain_stamps.diff, see tools/r5/dc_stamps.py; results wrong when a bit is set): boundary-step cycles
export SRGANFD_LIB=$PWD/sr_gan_fd_amd/libsrganfd_dcx.so SRGANFD_DC_STAMPS=1
for d in 0 1 2 3 4 7; do echo "SRGANFD_DC_DBG=$d"; SRGANFD_DC_DBG=$d timeout -k 10 120 python tools/r5/dc_stamps.py 4 128 128 2>&1 | grep -A1 "^boundary\|cycles from step" | grep -v "^--" | head -8; done
