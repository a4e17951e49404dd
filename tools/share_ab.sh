#!/bin/bash
export RBRT_HIP_LAB=1  # the scheduling knobs below are lab knobs (include/rbrt_hip_debug.h)
# A/B of the two kernel builds (with / without shared traversals in the drain) on the frame and on its 1/2, 1/4, 1/8
set -o pipefail
OUT=gpurun_out/share; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 > $OUT/test.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/test.log
RBRT_SHARE_BELOW=4000000000 timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 > $OUT/test_share.log 2>&1; echo "pytest(share everywhere) rc=$?"; tail -2 $OUT/test_share.log
for n in 1 2 4 8; do
  extra="--emulate-rank-of $n --isolated-steps 0"; [ $n = 1 ] && extra="--isolated-steps 6"
  timeout -k 10 400 python3 tools/ab.py --rounds 2 --steps $((10 * n)) --extra "$extra" "RBRT_SHARE_BELOW=0" "RBRT_SHARE_BELOW=4000000000" "RBRT_SHARE_BELOW=4000000000 RBRT_SHARE_IDLE=4" > $OUT/ab_r$n.log 2>&1
  echo "== 1/$n of the frame"; tail -3 $OUT/ab_r$n.log
done
