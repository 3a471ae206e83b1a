#!/bin/bash
# round-3 session 11: streaming conv with sixteen waves per workgroup (each wave half the output channels): parity, then kbench 16 vs 8 waves vs conv_igemm
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp11; mkdir -p $O
cd $R
export SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so
SRGANFD_USE_STREAM=1 SRGANFD_STREAM_WS=2 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_generator_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
for ws in 2 1; do
  echo "== stream, WS=$ws"
  SRGANFD_STREAM_WS=$ws timeout -k 10 300 python tools/kbench.py --dtype f16 --modes 8 --igv 2048 --dbg 0,17 --rounds 4 --reps 20 > $O/kbench_ws$ws.txt 2>&1 || { tail -5 $O/kbench_ws$ws.txt; exit 1; }
  grep -v amdgpu $O/kbench_ws$ws.txt | cut -c1-150
done
echo "== conv_igemm"
timeout -k 10 300 python tools/kbench.py --dtype f16 --modes 8 --igv 0 --rounds 4 --reps 20 > $O/kbench_igemm.txt 2>&1
grep -v amdgpu $O/kbench_igemm.txt | cut -c1-150
