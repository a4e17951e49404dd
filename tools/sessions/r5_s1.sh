#!/bin/bash
# Round 5, session 1: the new create / one-shot paths. Builder costs over mesh sizes, the refine tests, the driver's bench line.
set -o pipefail
OUT=gpurun_out/r5/s1; mkdir -p $OUT
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n 5 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step smoke 300 python3 __graft_entry__.py smoke || exit 1
step refine 600 python3 -m pytest tests/test_bvh_refine.py tests/test_bvh_device.py -m gpu -q --timeout 300 -x
step selflaunch 600 python3 -m pytest tests/test_multi_rank_one_gpu.py -m gpu -q --timeout 600 -k "without_a_launcher"
step sweep 600 python3 tools/create_sweep.py
step bench20 500 python3 bench.py --steps 20 --warmup 5 --cpu-col-stride 0
O=$OUT/cli; mkdir -p $O
python3 -m rbrt_amd.standin $O/bunny.obj > /dev/null 2>&1
sed "s#obj_filepath: bunny.obj#obj_filepath: $O/bunny.obj#" scenes/example_scene.yaml > $O/scene.yaml
for t in a b c; do
  step cli_$t 120 rbrt_amd/bin/rbrt --config $O/scene.yaml -t $O/out_$t.png --report $O/rep_$t.json --height 768 --width 1024 --samples 50
done
echo "session done"
