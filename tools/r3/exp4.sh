#!/bin/bash
# round-3 session 4: persistent tiles with cross-tile prefetch -- parity tests, per-shape A/B in one process, training-step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp4; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_generator_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so python tools/kbench.py --dtype f16 --modes 8 --igv 0,512 --rounds 5 --reps 20 > $O/kbench_persist.txt 2>&1 || { tail -5 $O/kbench_persist.txt; exit 1; }
grep -v amdgpu $O/kbench_persist.txt | cut -c1-150
bash tools/r3/ab_libs.sh $R/gpurun_in/lib_a.so $R/sr_gan_fd_amd/libsrganfd_hip.so g_only 10
bash tools/r3/ab_libs.sh $R/gpurun_in/lib_a.so $R/sr_gan_fd_amd/libsrganfd_hip.so gan 6
