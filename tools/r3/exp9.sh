#!/bin/bash
# round-3 session 9: streaming conv (conv_stream.hip) -- parity tests, then kbench stream vs conv_igemm
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp9; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_generator_gpu.py tests/test_gan_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so timeout -k 10 300 python tools/kbench.py --dtype f16 --modes 8 --igv 0,2048 --rounds 5 --reps 20 > $O/kbench.txt 2>&1 || { tail -5 $O/kbench.txt; exit 1; }
grep -v amdgpu $O/kbench.txt | cut -c1-175
SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so timeout -k 10 400 python tools/kbench.py --dtype f16 --modes 8 --igv 0,2048 --rounds 4 --reps 10 --set gan > $O/kbench_gan.txt 2>&1 || { tail -5 $O/kbench_gan.txt; exit 1; }
grep -v amdgpu $O/kbench_gan.txt | cut -c1-175
