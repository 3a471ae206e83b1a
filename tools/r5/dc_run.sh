#!/bin/bash
# one GPU call for the dense-chain work: parity tests, the bench against the five launches, the in-kernel timeline
set -o pipefail
out=gpurun_out/r5_dc_$1
timeout -k 10 420 python -m pytest tests/test_dense_chain_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > ${out}_tests.txt; rc=$?
cat ${out}_tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/r5/dense_chain_bench.py 2>&1 | grep -v amdgpu.ids > ${out}_bench.txt || exit 1
cat ${out}_bench.txt
export SRGANFD_LIB=$PWD/sr_gan_fd_amd/libsrganfd_dcx.so SRGANFD_DC_STAMPS=1
(timeout -k 10 120 python tools/r5/dc_stamps.py 16 32 32 && timeout -k 10 120 python tools/r5/dc_stamps.py 4 128 128 && timeout -k 10 120 python tools/r5/dc_stamps.py 32 128 128) 2>&1 | grep -v amdgpu.ids > ${out}_stamps.txt
cat ${out}_stamps.txt
