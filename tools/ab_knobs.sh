#!/bin/bash
# A/B of lab-knob settings, ONE PROCESS PER SETTING (round 4: inside one process the first scene's pipelined leg came out
# 3-4 % faster than every later scene's, whatever the knobs -- tools/endsweep.py with several variants is biased towards its
# first one in that column). usage: [SWEEP_ARGS=..] [PASSES=2] tools/ab_knobs.sh TAG "knobs a" "knobs b" ...   ("-": no knob)
set -o pipefail
TAG=$1; shift
O=gpurun_out/r4; mkdir -p $O; : > $O/abk_$TAG.log
for pass in $(seq 1 ${PASSES:-2}); do
  for v in "$@"; do
    timeout -k 10 300 python3 tools/endsweep.py --rounds ${ROUNDS:-2} ${SWEEP_ARGS:-} "$v" 2>&1 | tail -1 | sed "s/^/pass $pass  /" >> $O/abk_$TAG.log || exit 1
  done
done
cat $O/abk_$TAG.log
