#!/bin/bash
# same-box comparison of several library builds on a training step: tools/r3/ab3.sh <workload> <steps> lib1 lib2 ...
R=$GRAFT_REPO_ROOT; wl=$1; st=$2; shift 2
for rep in 1 2 3; do
  for lib in "$@"; do
    SRGANFD_LIB=$R/$lib python $R/bench.py --workload $wl --steps $st --warmup 3 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $lib', d['ms_per_step'])"
  done
done
