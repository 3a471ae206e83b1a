#!/bin/bash
# GAN step with / without the compile-time epilogue kinds 9 (r1 + y2) and 12 (mask + y2), alternating in one gpurun call
for i in 1 2 3; do for v in 0 1; do
  echo -n "SRGANFD_Y2_KINDS=$v: "; SRGANFD_Y2_KINDS=$v python bench.py --workload gan --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done; done
