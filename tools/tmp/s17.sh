#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5/s17; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_helpers.py tests/test_bvh_refine.py -m gpu -q --timeout 300 -x > $OUT/t.log 2>&1; tail -1 $OUT/t.log
bash tools/profile.sh > $OUT/profile.log 2>&1; tail -1 $OUT/profile.log
