#!/bin/bash
# Build a variant of the HIP library for A/B runs (scripts/ab_bench.py):  scripts/build_variant.sh NAME [-DFLAG ...]
#   -> build/ab/libocc_NAME.so   (same flags as __graft_entry__.build())
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC -Iinclude -Ioccclusionenv_amd/csrc -Ioccl -Iocclusionenv_amd/csrc \
  -o build/ab/libocc_$name.so occlusionenv_amd/csrc/occ_kernels.hip "$@"
echo "built build/ab/libocc_$name.so $*"
