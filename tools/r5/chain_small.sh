#!/bin/bash
# the layer-persistent dense-block launch (tools/experiments/r4_chain.diff, variant library gpurun_in/libsrganfd_chain.so) at the
# reference-default shapes: batch 16 at 72x72 / 48x48 / 32x32, batch 8 at 60x60, and the BASELINE shape for comparison
export SRGANFD_LIB=$GRAFT_REPO_ROOT/gpurun_in/libsrganfd_chain.so
for s in "16 72 72" "16 48 48" "16 32 32" "8 60 60" "32 128 128"; do
  timeout -k 10 120 python tools/r5/chain_probe.py $s 2>&1 | grep -v "^concurrent" || echo "FAILED $s"
done
