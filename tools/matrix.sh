#!/bin/bash
# The parity files under the builder / queue settings a host may choose (run on the GPU box through gpurun):
#   tools/matrix.sh a   host builder, device builder, no background tree        tools/matrix.sh b   no spatial splits, one builder thread, four hardware queues
set -o pipefail
OUT=gpurun_out/r5/matrix; mkdir -p $OUT
T="python3 -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_full_size.py tests/test_bvh_refine.py -m gpu -q --timeout 600"
run() { local name=$1; shift; echo "== $name"; env "$@" timeout -k 10 500 $T > $OUT/$name.log 2>&1; local rc=$?; tail -n 1 $OUT/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; }
if [ "${1:-a}" = a ]; then
  run host RBRT_BVH_BUILDER=host; run device RBRT_BVH_BUILDER=device; run no_refine RBRT_BVH_REFINE=0
else
  run no_spatial RBRT_HIP_LAB=1 RBRT_BVH_SPATIAL=0; run one_thread RBRT_BVH_THREADS=1; run queues4 GPU_MAX_HW_QUEUES=4
fi
echo "matrix done"
