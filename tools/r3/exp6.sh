#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp6; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for lib in sr_gan_fd_amd/libsrganfd_hip.so gpurun_in/lib_wgloaders.so sr_gan_fd_amd/libsrganfd_hip.so gpurun_in/lib_wgloaders.so; do
  echo "== $lib"; SRGANFD_LIB=$R/$lib python tools/wgbench.py --dtype f16 --variants 3 --splits 0 2>&1 | grep -v amdgpu
done
bash tools/r3/ab3.sh g_only 10 gpurun_in/lib_wgloaders.so sr_gan_fd_amd/libsrganfd_hip.so
