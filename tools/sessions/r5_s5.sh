#!/bin/bash
# Round 5, session 5: traversals shared between lanes in the BULK of a launch too (lab: RBRT_DRAIN_MODE bit 2), by idle-lane threshold.
set -o pipefail
OUT=gpurun_out/r5/s5; mkdir -p $OUT
timeout -k 10 600 python3 tools/endsweep.py --rounds 3 --new-camera 1 "-" "RBRT_DRAIN_MODE=5" "RBRT_DRAIN_MODE=5 RBRT_SHARE_IDLE=12" "RBRT_DRAIN_MODE=5 RBRT_SHARE_IDLE=20" "RBRT_DRAIN_MODE=5 RBRT_SHARE_IDLE=28" > $OUT/share_bulk.log 2>&1
tail -8 $OUT/share_bulk.log | cut -c1-260
timeout -k 10 300 python3 tools/lane_use.py > $OUT/lane_use_default.txt 2>&1; tail -25 $OUT/lane_use_default.txt
RBRT_HIP_LAB=1 RBRT_DRAIN_MODE=5 RBRT_SHARE_IDLE=12 timeout -k 10 300 python3 tools/lane_use.py > $OUT/lane_use_share12.txt 2>&1; tail -25 $OUT/lane_use_share12.txt
echo "session done"
