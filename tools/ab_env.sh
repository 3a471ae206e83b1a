# usage: VAR=SRGANFD_SN_BATCH WL=gan bash gpurun_in/ab_env.sh   (alternates VAR=1 / VAR=0 on the same box)
for v in ${VALS:-1 0 1 0 1 0}; do
  export ${VAR}=$v
  python bench.py --workload ${WL:-gan} --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'])" || exit 1
done
