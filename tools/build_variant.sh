#!/bin/bash
# Builds a variant of the product library for A/B runs: tools/build_variant.sh NAME [extra hipcc flags...]
#   -> rbrt_amd/lib/variants/librbrt_hip_NAME.so   (select it with RBRT_HIP_LIB=..., see tools/ab.py)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p rbrt_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" -shared \
    -o rbrt_amd/lib/variants/librbrt_hip_$NAME.so rbrt_amd/csrc/kernels.hip rbrt_amd/csrc/bvh_device.hip rbrt_amd/csrc/api.cpp rbrt_amd/csrc/bvh.cpp
echo "built rbrt_amd/lib/variants/librbrt_hip_$NAME.so"
