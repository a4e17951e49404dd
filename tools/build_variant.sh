#!/bin/bash
# Builds a variant of the product library for A/B runs: tools/build_variant.sh NAME [extra hipcc flags...]
#   -> rbrt_amd/lib/variants/librbrt_hip_NAME.so   (select it with RBRT_HIP_LIB=..., see tools/ab.py)
# The flags are the Makefile's (make -s print-hipflags) plus the extra ones.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p rbrt_amd/lib/variants
/opt/rocm/bin/hipcc $(make -s print-hipflags) "$@" -shared \
    -o rbrt_amd/lib/variants/librbrt_hip_$NAME.so rbrt_amd/csrc/kernels.hip rbrt_amd/csrc/bvh_device.hip rbrt_amd/csrc/api.cpp rbrt_amd/csrc/bvh.cpp
echo "built rbrt_amd/lib/variants/librbrt_hip_$NAME.so"
