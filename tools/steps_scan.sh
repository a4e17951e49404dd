#!/bin/bash
export RBRT_HIP_LAB=1  # the scheduling knobs below are lab knobs (include/rbrt_hip_debug.h)
# ms/step of the pipelined leg against the number of timed steps (the first launch of a stream of frames finds the GPU idle)
run() { local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env "${envs[@]}" timeout -k 10 120 python3 bench.py --cpu-col-stride 0 --isolated-steps 0 --single-frames 0 "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])")
  echo "${envs[*]} $* -> $r"; }
for st in 5 10 20 40; do
  run X=1 -- --steps $st --warmup 2
  run X=1 -- --steps $st --warmup 0
  run RBRT_WAVES_PER_CU=8 RBRT_WORK_STRIPES=0 -- --steps $st --warmup 2 --pipeline 3
done
