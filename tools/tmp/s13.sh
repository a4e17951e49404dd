#!/bin/bash
set -o pipefail
OUT=gpurun_out/r5/s13; mkdir -p $OUT
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1; local rc=$?; echo "rc=$rc $(tail -1 $OUT/$name.log | cut -c1-150)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step smoke 300 python3 __graft_entry__.py smoke || exit 1
step pytest 1100 python3 -m pytest tests -m gpu -q --timeout 600 || exit 1
step shares2 500 python3 bench.py --steps 40 --warmup 5 --cpu-col-stride 0 --emulate-all 2,4,8
step shares3 900 python3 bench.py --steps 6 --warmup 1 --cpu-col-stride 0 --width 1920 --height 1080 --spp 512 --emulate-all 2,4,8
for i in 1 2; do step driver_$i 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-col-stride 0; done
step c100 400 python3 bench.py --steps 100 --warmup 5 --cpu-col-stride 0 --one-shot 0 --single-frames 3
for f in shares2 shares3; do grep -v amdgpu $OUT/$f.log | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['whole_frame_ms']); [print(k, v['ms_per_rank'], v['max_over_mean'], v['predicted_scaling']) for k,v in j['worlds'].items()]"; done
python3 - <<'PY'
import json
for f in ('driver_1','driver_2','c100'):
    j=json.loads([l for l in open(f'gpurun_out/r5/s13/{f}.log') if l.startswith('{')][-1]); r=j['roofline']
    print(f, j['value'], j['ms_per_step'], 'iso', r['kernel_ms'], 'single', j['single_frame']['ms'], 'oneshot', (j.get('one_shot') or {}).get('ms'), j['config']['image_sha256_16'])
PY
