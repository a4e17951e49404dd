#!/bin/bash
# Round 5, session 4: the mesh shortcut (a mesh no triangle of which a ray can hit is not walked): config 4 with and without, and what
# having the code costs the headline frame (A/B of library builds, one process per build).
set -o pipefail
OUT=gpurun_out/r5/s4; mkdir -p $OUT
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n 3 "$OUT/$name.log" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step smoke 300 python3 __graft_entry__.py smoke || exit 1
step parity 900 python3 -m pytest tests/test_full_size.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x || exit 1
RDIR=r5/s4 PASSES=3 ROUNDS=3 step ab 900 bash tools/ab_variants.sh shortcut noshortcut
B="python3 bench.py --steps 100 --warmup 5 --cpu-col-stride 0 --single-frames 3 --one-shot 0 --same-camera-steps 0"
step b4 300 $B --config 4
RBRT_HIP_LIB=rbrt_amd/lib/variants/librbrt_hip_noshortcut.so step b4_no 300 $B --config 4
python3 - <<'PY'
import json
for f in ('b4','b4_no'):
    j=json.loads([l for l in open(f'gpurun_out/r5/s4/{f}.log') if l.startswith('{')][-1]); r=j['roofline']
    print(f, 'value', j['value'], 'ms', j['ms_per_step'], 'iso', r['kernel_ms'], 'nodes/s', r['per_sample']['nodes_visited'], 'single', j.get('single_frame',{}).get('ms'), j['config']['image_sha256_16'])
PY
echo "session done"
