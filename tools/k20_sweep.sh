#!/bin/bash
# Grid size of an overlapped launch against the length of the stream (GPU box): ms per step for 20 and 100 steps, frame /
# half / quarter / eighth of config 2, waves per CU 3..8 (lab knob; 0 = the library's rule).
export RBRT_HIP_LAB=1
for n in 1 2 4 8; do a=""; [ $n != 1 ] && a="--emulate-rank-of $n"
  for w in ${WAVES:-0 3 4 6 8}; do
    line="share 1/$n waves $w:"
    for k in 20 100; do
      v=$(RBRT_OVERLAP_WAVES_PER_CU=$w timeout -k 10 200 python3 bench.py --cpu-col-stride 0 --isolated-steps 0 --single-frames 0 --same-camera-steps 0 --warmup 5 --steps $k $a 2>/dev/null | grep "^{" | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'])")
      line="$line  K=$k $v"
    done
    echo "$line"
  done
done
