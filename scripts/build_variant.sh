#!/bin/bash
# Build one variant of the HIP library for A/B runs:  bash scripts/build_variant.sh NAME [-Dflags...]  -> build/ab/libocc_NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 ${SLP:--fno-slp-vectorize} -shared -fPIC -Iinclude -Iocclusionenv_amd/csrc "$@" \
  -o build/ab/libocc_$name.so occlusionenv_amd/csrc/occ_kernels.hip
echo "built build/ab/libocc_$name.so"
