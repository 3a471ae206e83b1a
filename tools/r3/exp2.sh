#!/bin/bash
# round-3 session 2: compile-time epilogue kinds -- parity tests, per-shape A/B in one process, training-step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_exp2; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_generator_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so python tools/kbench.py --dtype f16 --modes 8 --igv 0,256 --rounds 5 --reps 20 > $O/kbench_kinds.txt 2>&1 || { tail -5 $O/kbench_kinds.txt; exit 1; }
grep -v amdgpu $O/kbench_kinds.txt | cut -c1-150
for v in 0 1 0 1; do
  SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so SRGANFD_NO_EPI_KINDS=$v python bench.py --workload g_only --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('g_only no_kinds=$v', d['ms_per_step'])"
done
for v in 0 1 0 1; do
  SRGANFD_LIB=$R/build_exp/libsrganfd_exp.so SRGANFD_NO_EPI_KINDS=$v python bench.py --workload gan --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gan no_kinds=$v', d['ms_per_step'])"
done
