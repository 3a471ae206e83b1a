#!/bin/bash
# round-3 session 3: 16-bit transposed epilogue tile (kind 0) -- bitwise test, step A/B against the previous build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp3; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "epilogue_kinds or conv2d_forward" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
bash tools/r3/ab_libs.sh $R/gpurun_in/lib_a.so $R/sr_gan_fd_amd/libsrganfd_hip.so g_only 10
SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so python tools/kbench.py --dtype f16 --modes 8 --igv 0,256 --rounds 5 --reps 20 --only fwd > $O/kbench_tile16.txt 2>&1 || { tail -5 $O/kbench_tile16.txt; exit 1; }
grep -v amdgpu $O/kbench_tile16.txt | cut -c1-150
