#!/bin/bash
# same-box A/B of two library builds on a training step: tools/r3/ab_libs.sh <old.so> <new.so> [workload] [steps]
R=$GRAFT_REPO_ROOT; old=$1; new=$2; wl=${3:-g_only}; st=${4:-10}
for v in new old new old new old; do
  if [ $v = old ]; then lib=$old; else lib=$new; fi
  SRGANFD_LIB=$lib python $R/bench.py --workload $wl --steps $st --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $v', d['ms_per_step'])"
done
