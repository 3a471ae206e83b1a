# Same-box A/B of a training step on the GPU box: alternates the in-tree library ("new") with a baseline build at gpurun_in/libold.so ("old").
#   WL=gan bash tools/ab_step.sh
R=$GRAFT_REPO_ROOT
for v in new old new old new old; do
  if [ $v = old ]; then export SRGANFD_LIB=$R/gpurun_in/libold.so; else unset SRGANFD_LIB; fi
  python bench.py --workload ${WL:-g_only} --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
done
