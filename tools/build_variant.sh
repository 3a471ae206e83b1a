#!/bin/bash
# Build an A/B variant of the HIP library: tools/build_variant.sh <out.so> <conv_igemm source> [extra hipcc flags]
# (run a benchmark against it with SRGANFD_LIB=<out.so>)
set -e
out=$1; conv=$2; shift 2
src=$(dirname "$0")/../sr-gan-fd_amd/csrc
others=$(ls "$src"/*.hip | grep -v conv_igemm.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$src" "$@" -o "$out" "$conv" $others
