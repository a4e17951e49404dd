#!/bin/bash
# Round 4 closing session: the gpu tests, the default bench line, configs 2r / 4 / 4v, the eighth, then the rocprofv3 passes (tools/profile.sh).
set -o pipefail
O=gpurun_out/r4; mkdir -p $O
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$O/z_$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n ${TAILN:-2} "$O/z_$name.log" | cut -c1-500
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step smoke 300 python3 __graft_entry__.py smoke || exit 1
step pytest 1100 python3 -m pytest tests -m gpu -q -x --timeout 600 || exit 1
step bench 500 python3 bench.py --check
C="--cpu-col-stride 0"
step bench_r8 300 python3 bench.py $C --emulate-rank-of 8
step bench_r4 300 python3 bench.py $C --emulate-rank-of 4
step bench_r2 300 python3 bench.py $C --emulate-rank-of 2
step bench2r 300 python3 bench.py $C --config 2r
step bench4 300 python3 bench.py $C --config 4
step bench4v 300 python3 bench.py $C --config 4v
step moving 300 python3 tools/moving_camera.py
step lane_use 300 python3 tools/lane_use.py
step lane_use_r8 300 python3 tools/lane_use.py --world 8
if [ -f rbrt_amd/lib/librbrt_hip_timers.so ]; then  # (make timers, before the session: the box has no need to compile)
  RBRT_HIP_LIB=rbrt_amd/lib/librbrt_hip_timers.so step regions_full 300 python3 tools/region_profile.py
  RBRT_HIP_LIB=rbrt_amd/lib/librbrt_hip_timers.so step regions_r8 300 python3 tools/region_profile.py --emulate-rank-of 8
fi
step profile 1100 bash tools/profile.sh
echo "session done"
