#!/bin/bash
# EXPERIMENT: dense-block weight gradient (kernel + slab reduction) over the pixel-split count at the reference's default shapes
for cfg in "16 32" "16 48" "8 60" "16 72"; do set -- $cfg
  echo "== batch $1, $2 x $2"; timeout -k 10 200 python tools/wgbench.py --dtype f16 --batch $1 --size $2 --variants 0 --splits 0,36,18,12,9,6,4,2 2>&1 | grep -v amdgpu.ids | tail -9
done
