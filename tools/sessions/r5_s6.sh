#!/bin/bash
# Round 5, session 6: elastic launches (helper launches by a watcher thread). Tests, the forced mode over the parity suite, and
# the stream's ends: bench at 20 / 40 / 100 steps with and without (RBRT_HELPERS=0), alternating.
set -o pipefail
OUT=gpurun_out/r5/s6; mkdir -p $OUT
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n 3 "$OUT/$name.log" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step smoke 300 python3 __graft_entry__.py smoke || exit 1
step helpers 600 python3 -m pytest tests/test_helpers.py -m gpu -q --timeout 300 -x || exit 1
RBRT_HIP_LAB=1 RBRT_HELPERS=2 RBRT_POISON_SAMPLES=1 step forced_suite 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_golden.py tests/test_primary_cull.py tests/test_multi_rank_one_gpu.py -m gpu -q --timeout 600 -x || exit 1
step pytest 1100 python3 -m pytest tests -m gpu -q --timeout 600 -x || exit 1
B="python3 bench.py --warmup 5 --cpu-col-stride 0 --single-frames 3 --one-shot 0 --isolated-steps 4"
for pass in 1 2 3; do
  for k in 20 100; do
    step on_${k}_$pass 300 $B --steps $k
    RBRT_HIP_LAB=1 RBRT_HELPERS=0 step off_${k}_$pass 300 $B --steps $k
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/s6/o*_*.log')):
    j=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], 'ms', j['ms_per_step'], 'same', j['ms_per_step_same_camera'], 'helpers', j['config'].get('helper_launches_timed_region'), 'single', j['single_frame']['ms'], j['config']['image_sha256_16'])
PY
echo "session done"
