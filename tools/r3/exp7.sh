#!/bin/bash
# round-3 session 7: balanced pixel splits per channel group in the weight-gradient launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp7; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_generator_gpu.py -x -q -m gpu -k "wgrad or g_only or block" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
echo "== old (uniform splits)"; SRGANFD_LIB=$R/gpurun_in/lib_b.so python tools/wgbench.py --dtype f16 --variants 3 --splits 0 2>&1 | grep -v amdgpu
for b in 0.0 0.12 0.25 0.4 0.6; do echo "== balanced, bias $b"; SRGANFD_WG_LIGHT_BIAS=$b python tools/wgbench.py --dtype f16 --variants 3 --splits 0 2>&1 | grep -v amdgpu; done
echo "== old (uniform splits)"; SRGANFD_LIB=$R/gpurun_in/lib_b.so python tools/wgbench.py --dtype f16 --variants 3 --splits 0 2>&1 | grep -v amdgpu
bash tools/r3/ab3.sh g_only 10 gpurun_in/lib_b.so sr_gan_fd_amd/libsrganfd_hip.so
