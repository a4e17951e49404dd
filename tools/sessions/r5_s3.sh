#!/bin/bash
# Round 5, session 3: spatial splits in the host builder (configs 2, 2r, 4v, each against RBRT_BVH_SPATIAL=0 on the same box), the suite, cold start.
set -o pipefail
OUT=gpurun_out/r5/s3; mkdir -p $OUT
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n 3 "$OUT/$name.log" | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step smoke 300 python3 __graft_entry__.py smoke || exit 1
step pytest 1100 python3 -m pytest tests -m gpu -q --timeout 600 -x || exit 1
B="python3 bench.py --steps 100 --warmup 5 --cpu-col-stride 0 --single-frames 3 --one-shot 3 --same-camera-steps 0"
for cfg in 2 2r 4v; do
  step b_${cfg}_sbvh 400 $B --config $cfg
  RBRT_HIP_LAB=1 RBRT_BVH_SPATIAL=0 step b_${cfg}_nosplit 400 $B --config $cfg
done
step bench20 500 python3 bench.py --steps 20 --warmup 5 --cpu-col-stride 0
O=$OUT/cli; mkdir -p $O
python3 -m rbrt_amd.standin $O/bunny.obj > /dev/null 2>&1
sed "s#obj_filepath: bunny.obj#obj_filepath: $O/bunny.obj#" scenes/example_scene.yaml > $O/scene.yaml
for t in a b; do
  step cli_$t 120 rbrt_amd/bin/rbrt --config $O/scene.yaml -t $O/out_$t.png --report $O/rep_$t.json --height 768 --width 1024 --samples 50
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/s3/b_*.log')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'no line'); continue
    r=j['roofline']
    print(f.split('/')[-1], 'value', j['value'], 'ms', j['ms_per_step'], 'iso', r['kernel_ms'], 'nodes/s', r['per_sample']['nodes_visited'], 'tris/s', r['per_sample']['tris_tested'], 'single', j.get('single_frame',{}).get('ms'), 'oneshot', j.get('one_shot',{}).get('ms'), r['lds_stack']['bvh_builder'][:60], 'nodes', r['lds_stack']['bvh_nodes'], 'setup', j['config']['setup_s_excluded'])
for f in sorted(glob.glob('gpurun_out/r5/s3/cli/rep_*.json')):
    print(f, open(f).read()[-420:])
PY
echo "session done"
