#!/bin/bash
# round-3 session 8: 16-wave double-buffered 64-channel conv tiles
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp8; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_generator_gpu.py tests/test_gan_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so python tools/kbench.py --dtype f16 --modes 8 --igv 0,1024 --rounds 5 --reps 20 --only "64" > $O/kbench_db.txt 2>&1 || { tail -5 $O/kbench_db.txt; exit 1; }
grep -v amdgpu $O/kbench_db.txt | cut -c1-175
SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so python tools/kbench.py --dtype f16 --modes 8 --igv 0,1024 --rounds 4 --reps 10 --set gan > $O/kbench_db_gan.txt 2>&1 || { tail -5 $O/kbench_db_gan.txt; exit 1; }
grep -v amdgpu $O/kbench_db_gan.txt | cut -c1-175
bash tools/r3/ab3.sh g_only 10 gpurun_in/lib_c.so sr_gan_fd_amd/libsrganfd_hip.so
bash tools/r3/ab3.sh gan 6 gpurun_in/lib_c.so sr_gan_fd_amd/libsrganfd_hip.so
