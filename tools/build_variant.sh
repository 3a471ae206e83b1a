#!/bin/bash
# Build an A/B variant of the HIP library: tools/build_variant.sh <out.so> <conv_igemm source> [extra hipcc flags]
# (run a benchmark against it with SRGANFD_LIB=<out.so>); -DSRGANFD_EXPERIMENT adds the debug switches and the experiment kernels (ring, conv-pair fusion, LDS-DMA streaming conv)
set -e
out=$1; conv=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/sr_gan_fd_amd/csrc
others=$(ls "$src"/*.hip | grep -v conv_igemm.hip)
case " $* " in *SRGANFD_EXPERIMENT*) others="$others $root/tools/experiments/conv3x3_ring.hip $root/tools/experiments/conv_pair.hip $root/tools/experiments/conv_stream.hip";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$src" "$@" -o "$out" "$conv" $others
