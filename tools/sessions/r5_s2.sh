#!/bin/bash
# Round 5, session 2: the whole GPU suite on the new create path, every rank's share (configs 2 and 3), create sweep, CLI report.
set -o pipefail
OUT=gpurun_out/r5/s2; mkdir -p $OUT
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n 4 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step smoke 300 python3 __graft_entry__.py smoke || exit 1
step refine 600 python3 -m pytest tests/test_bvh_refine.py -m gpu -q --timeout 300 -x || exit 1
step pytest 1100 python3 -m pytest tests -m gpu -q --timeout 600
step sweep 600 python3 tools/create_sweep.py
step shares2 500 python3 bench.py --steps 40 --warmup 5 --cpu-col-stride 0 --emulate-all 2,4,8
step shares3 900 python3 bench.py --steps 6 --warmup 1 --cpu-col-stride 0 --width 1920 --height 1080 --spp 512 --emulate-all 2,4,8
RBRT_HIP_LAB=1 RBRT_TRACE_CREATE=1 step bench20 500 python3 bench.py --steps 20 --warmup 5 --cpu-col-stride 0
O=$OUT/cli; mkdir -p $O
python3 -m rbrt_amd.standin $O/bunny.obj > /dev/null 2>&1
sed "s#obj_filepath: bunny.obj#obj_filepath: $O/bunny.obj#" scenes/example_scene.yaml > $O/scene.yaml
for t in a b; do
  RBRT_HIP_LAB=1 RBRT_TRACE_CREATE=1 step cli_$t 120 rbrt_amd/bin/rbrt --config $O/scene.yaml -t $O/out_$t.png --report $O/rep_$t.json --height 768 --width 1024 --samples 50
done
echo "session done"
