#!/bin/bash
# One GPU-box session that makes round 5's records (run from the repo root through gpurun; tools/profile.sh is its own session):
#   tests (the suite; the parity files once more with a helper launch behind EVERY launch and poisoned sample buffers; the
#   tile pass against 500 fuzzed cameras), the driver's bench line three times, 100-step lines of configs 2 / 2r / 4 / 4v,
#   elastic launches on / off at 20 steps (frame and eighth), every rank's share (configs 2 and 3), the create sweep, the CLI.
# gpurun limits a call to 20 minutes: tools/r5_final.sh tests | bench | ends | shares | rest  runs one part (default: all).
# A step that TIMES OUT ends the session (a hung kernel must not be followed by more GPU work); one that fails does not.
set -o pipefail
OUT=gpurun_out/r5/final; mkdir -p $OUT
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n 2 "$OUT/$name.log" | cut -c1-200
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
PART=${1:-all}
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
if want tests; then
step smoke 300 python3 __graft_entry__.py smoke || exit 1
step pytest 1100 python3 -m pytest tests -m gpu -q --timeout 600
RBRT_HIP_LAB=1 RBRT_HELPERS=2 RBRT_POISON_SAMPLES=1 step forced_helpers 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_golden.py tests/test_multi_rank_one_gpu.py -m gpu -q --timeout 600
RBRT_FUZZ_CAMERAS=500 RBRT_FUZZ_SCENES=150 step fuzz500 1100 python3 -m pytest tests/test_primary_cull.py tests/test_gpu_parity.py -m gpu -q --timeout 1000 -k "fuzz or random"
fi
if want bench; then
for i in 1 2 3; do step driver_$i 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-col-stride 0; done
B="python3 bench.py --steps 100 --warmup 5 --cpu-col-stride 0 --single-frames 3 --one-shot 3"
for cfg in 2 2r 4 4v; do step cfg_$cfg 400 $B --config $cfg; done
fi
if want ends; then
E="python3 bench.py --warmup 5 --cpu-col-stride 0 --single-frames 0 --one-shot 0 --isolated-steps 0 --same-camera-steps 0 --steps 20"
for pass in 1 2 3; do
  step frame_on_$pass 300 $E; RBRT_HIP_LAB=1 RBRT_HELPERS=0 step frame_off_$pass 300 $E
  step eighth_on_$pass 300 $E --emulate-rank-of 8; RBRT_HIP_LAB=1 RBRT_HELPERS=0 step eighth_off_$pass 300 $E --emulate-rank-of 8
done
fi
if want shares; then
step shares2 500 python3 bench.py --steps 40 --warmup 5 --cpu-col-stride 0 --emulate-all 2,4,8
step shares3 900 python3 bench.py --steps 6 --warmup 1 --cpu-col-stride 0 --width 1920 --height 1080 --spp 512 --emulate-all 2,4,8
fi
if want rest; then
step sweep 600 python3 tools/create_sweep.py
O=$OUT/cli; mkdir -p $O
python3 -m rbrt_amd.standin $O/bunny.obj > /dev/null 2>&1
sed "s#obj_filepath: bunny.obj#obj_filepath: $O/bunny.obj#" scenes/example_scene.yaml > $O/scene.yaml
for t in a b; do step cli_$t 120 rbrt_amd/bin/rbrt --config $O/scene.yaml -t $O/out_$t.png --report $O/rep_$t.json --height 768 --width 1024 --samples 50; done
fi
python3 - <<'PY'
import json, glob, os
def last(f):
    try: return json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception: return None
for f in sorted(glob.glob('gpurun_out/r5/final/*.log')):
    j = last(f)
    if not j or 'value' not in j: continue
    r = j['roofline']
    print(os.path.basename(f)[:-4], 'value', j['value'], 'ms', j['ms_per_step'], 'iso', r.get('kernel_ms'), 'frac', r.get('frac'), 'helpers', j['config'].get('helper_launches_timed_region'),
          'single', (j.get('single_frame') or {}).get('ms'), 'one_shot', (j.get('one_shot') or {}).get('ms'), j['config']['image_sha256_16'])
PY
echo "session done"
