#!/bin/bash
export RBRT_HIP_LAB=1  # the scheduling knobs below are lab knobs (include/rbrt_hip_debug.h)
# The full frame under scheduling variants (pipeline depth x grid size)
run() { local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  r=$(env "${envs[@]}" timeout -k 10 120 python3 bench.py --steps 16 --warmup 4 --cpu-col-stride 0 --isolated-steps 0 --single-frames 0 "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])")
  echo "${envs[*]} $* -> $r"; }
for rep in 1 2; do
run X=1 --
run RBRT_WAVES_PER_CU=8 -- --pipeline 3
run RBRT_WAVES_PER_CU=8 -- --pipeline 2
run RBRT_WAVES_PER_CU=7 -- --pipeline 3
run RBRT_WAVES_PER_CU=6 -- --pipeline 3
run RBRT_WAVES_PER_CU=5 -- --pipeline 4
run RBRT_WAVES_PER_CU=4 -- --pipeline 5
run RBRT_WAVES_PER_CU=4 -- --pipeline 4
run RBRT_WAVES_PER_CU=8 RBRT_WORK_STRIPES=0 -- --pipeline 3
run RBRT_WAVES_PER_CU=8 RBRT_Y_LOW=32 -- --pipeline 3
run RBRT_WAVES_PER_CU=8 RBRT_SHARE_IDLE=16 -- --pipeline 3
done
