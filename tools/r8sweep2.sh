#!/bin/bash
export RBRT_HIP_LAB=1  # the scheduling knobs below are lab knobs (include/rbrt_hip_debug.h)
# An eighth of the frame (rank 0 of 8) under scheduling variants: grid size x pipeline depth, drain mode, watermarks
mkdir -p gpurun_out
run() { # env... -- extra bench args
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env "${envs[@]}" timeout -k 10 120 python3 bench.py --steps 60 --warmup 8 --cpu-col-stride 0 --isolated-steps 0 --single-frames 0 --emulate-rank-of 8 "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])")
  echo "${envs[*]} $* -> $r"
}
for rep in 1 2; do
run X=1 --
run RBRT_WAVES_PER_CU=8 -- --pipeline 3
run RBRT_WAVES_PER_CU=8 -- --pipeline 4
run RBRT_WAVES_PER_CU=6 -- --pipeline 4
run RBRT_WAVES_PER_CU=4 -- --pipeline 4
run RBRT_WAVES_PER_CU=4 -- --pipeline 6
run RBRT_WAVES_PER_CU=16 -- --pipeline 2
run RBRT_DRAIN_MODE=0 --
run RBRT_DRAIN_MODE=3 --
run RBRT_Y_LOW=20 --
run RBRT_Y_LOW=36 --
run RBRT_SHARE_IDLE=1 --
run RBRT_SHARE_IDLE=8 --
run RBRT_WORK_STRIPES_OVERLAP=16 --
done
