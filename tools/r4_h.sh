#!/bin/bash
# Round 4 session h: the list of an ISOLATED launch (RBRT_TILE_ISOLATED_MODE: 0 row-major, 1 heavy first, 4 row-major with a light tail) on six workloads
set -o pipefail
O=gpurun_out/r4; mkdir -p $O
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; timeout -k 10 "$tmo" "$@" > "$O/h_$name.log" 2>&1; local rc=$?; echo "== $name rc=$rc"; sed -n '/summary/,$p' "$O/h_$name.log" | cut -c1-260
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step pytest 900 python3 -m pytest tests -m gpu -q -x --timeout 600 -k "parity or cull or full_size or multi_rank" ; tail -2 $O/h_pytest.log
V='RBRT_TILE_ISOLATED_MODE=0 RBRT_TILE_ISOLATED_MODE=1 RBRT_TILE_ISOLATED_MODE=4 RBRT_TILE_ISOLATED_MODE=4@RBRT_TILE_TAIL_DIV=16 RBRT_TILE_ISOLATED_MODE=4@RBRT_TILE_TAIL_DIV=4 RBRT_TILE_ISOLATED_MODE=4@RBRT_TILE_TAIL_DIV=32'
VV=(); for v in $V; do VV+=("${v//@/ }"); done
step cfg2 400 python3 tools/endsweep.py --rounds 3 --worlds 1,8 "${VV[@]}"
step rough 400 python3 tools/endsweep.py --rounds 3 --worlds 1 --mesh rough "${VV[@]}"
step cfg4 400 python3 tools/endsweep.py --rounds 3 --worlds 1 --triangles 871414 "${VV[@]}"
step cfg4v 400 python3 tools/endsweep.py --rounds 3 --worlds 1 --triangles 871414 --mesh-scale 450 --mesh-translation 50,-18,-145 "${VV[@]}"
step header 400 python3 tools/endsweep.py --rounds 3 --worlds 1 --scene scenes/header_card.yaml --width 1024 --height 1024 --spp 32 "${VV[@]}"
echo "session done"
