#!/bin/bash
# A/B of library variants built by tools/build_variant.sh: tools/ab_libs.sh TAG name1 name2 ...   (on the GPU box)
TAG=$1; shift
V=()
for n in "$@"; do V+=("RBRT_HIP_LIB=rbrt_amd/lib/variants/librbrt_hip_$n.so"); done
mkdir -p gpurun_out/r3
timeout -k 10 1000 python3 tools/ab.py --rounds ${ROUNDS:-3} --steps ${STEPS:-20} --isolated-steps ${ISO:-6} "${V[@]}" > gpurun_out/r3/ab_$TAG.log 2>&1
tail -$(( ${#V[@]} + 1 )) gpurun_out/r3/ab_$TAG.log
