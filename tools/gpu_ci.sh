#!/bin/bash
# One GPU-box session: smoke -> GPU tests -> bench (+ a counting run). A step that TIMES OUT ends the session
# (a hung kernel must not be followed by more GPU work); a step that merely fails does not.
# usage: tools/gpu_ci.sh <tag> [pytest -k expression]
set -o pipefail
TAG=${1:-run}
KEXPR=${2:-}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
step() { # name timeout cmd...
  local name=$1 tmo=$2; shift 2
  echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n 6 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
  return $rc
}
step smoke 300 python3 __graft_entry__.py smoke || exit 1
if [ -n "$KEXPR" ]; then
  step pytest 1000 python3 -m pytest tests -m gpu -q --timeout 600 -k "$KEXPR"
else
  step pytest 1000 python3 -m pytest tests -m gpu -q --timeout 600
fi
step bench 400 python3 bench.py --steps 20 --warmup 3 --check
cp "$OUT/bench.log" "$OUT/bench.json" 2>/dev/null
RBRT_BENCH_DEBUG=1 step bench_dbg 300 python3 bench.py --steps 3 --warmup 1 --cpu-col-stride 0 --isolated-steps 0 --single-frames 0
step bench_r8 300 python3 bench.py --steps 40 --warmup 5 --cpu-col-stride 0 --emulate-rank-of 8
echo "session done"
