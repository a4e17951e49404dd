#!/bin/bash
# Round 4 session f: tests, bench lines for configs 2 / 2r / 4 / 4v and the eighth, the work-list order on five workloads.
set -o pipefail
O=gpurun_out/r4; mkdir -p $O
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$O/f_$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n ${TAILN:-3} "$O/f_$name.log" | cut -c1-600
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step pytest 1100 python3 -m pytest tests -m gpu -q -x --timeout 600 || exit 1
C="--cpu-col-stride 0"
step bench2 300 python3 bench.py $C
step bench_r8 300 python3 bench.py $C --steps 40 --warmup 5 --emulate-rank-of 8
step bench2r 300 python3 bench.py $C --config 2r
step bench4 300 python3 bench.py $C --config 4 --single-frames 0
step bench4v 300 python3 bench.py $C --config 4v --single-frames 0
TAILN=6
V='- RBRT_TILE_CLASSES=2 RBRT_TILE_CLASSES=1 RBRT_TILE_CLASSES=3 RBRT_TILE_ORDER=1'
step order_cfg2 400 python3 tools/endsweep.py --rounds 3 --new-camera 1 --worlds 1,8 $V
step order_rough 400 python3 tools/endsweep.py --rounds 3 --new-camera 1 --worlds 1 --mesh rough $V
step order_cfg4 400 python3 tools/endsweep.py --rounds 3 --new-camera 1 --worlds 1 --triangles 871414 $V
step order_cfg4v 400 python3 tools/endsweep.py --rounds 3 --new-camera 1 --worlds 1 --triangles 871414 --mesh-scale 450 --mesh-translation 50,-18,-145 $V
step order_header 400 python3 tools/endsweep.py --rounds 3 --new-camera 1 --worlds 1 --scene scenes/header_card.yaml --width 1024 --height 1024 --spp 32 $V
echo "session done"
