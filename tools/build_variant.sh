#!/bin/bash
# Build the round-3 EXPERIMENT library (debug ablation switches, s_memrealtime stamps, 32x32x16 forms, ring / stream / chain / pair-fusion
# kernels; ABI version 3) from the frozen sources under tools/experiments/r3_src -- the product sources under sr_gan_fd_amd/csrc carry none
# of it since round 4.  Run benchmarks against it with SRGANFD_LIB=<out.so> on the round-3 Python tree (git tag / commit 8e2630f).
#   tools/build_variant.sh <out.so> [conv_igemm source] [extra hipcc flags]
set -e
out=$1; conv=${2:-}; shift; [ $# -gt 0 ] && shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/tools/experiments/r3_src
[ -z "$conv" ] && conv=$src/conv_igemm.hip
prod=$root/sr_gan_fd_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSRGANFD_EXPERIMENT -I"$src" "$@" -o "$out" "$conv" "$src"/wgrad.hip "$src"/abi.hip "$src"/elementwise.hip "$src"/pack.hip "$src"/degrade.hip \
  "$root"/tools/experiments/conv3x3_ring.hip "$root"/tools/experiments/conv_pair.hip "$root"/tools/experiments/conv_stream.hip
