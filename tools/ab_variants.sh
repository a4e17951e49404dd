#!/bin/bash
# A/B of library builds on the GPU box: the product library against variants (tools/build_variant.sh), each in its own
# process (tools/endsweep.py, default knobs), alternating for PASSES passes. usage: tools/ab_variants.sh TAG name1 [name2 ...]
set -o pipefail
TAG=$1; shift
O=gpurun_out/${RDIR:-r5}; mkdir -p $O; : > $O/ab_$TAG.log
for pass in $(seq 1 ${PASSES:-2}); do
  for n in product "$@"; do
    lib=rbrt_amd/lib/librbrt_hip.so; [ "$n" != product ] && lib=rbrt_amd/lib/variants/librbrt_hip_$n.so
    echo "## pass $pass lib $n" >> $O/ab_$TAG.log
    RBRT_HIP_LIB=$lib timeout -k 10 300 python3 tools/endsweep.py --rounds ${ROUNDS:-3} ${SWEEP_ARGS:-} "-" 2>&1 | tail -1 >> $O/ab_$TAG.log || exit 1
  done
done
cat $O/ab_$TAG.log
