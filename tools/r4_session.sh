#!/bin/bash
# Round 4: one GPU-box session = the gpu tests, the default bench line, and a kernel trace of a stream of eighths.
# A step that TIMES OUT ends the session. usage: tools/r4_session.sh TAG [pytest -k expression]
set -o pipefail
TAG=${1:-s}; KEXPR=${2:-}
O=gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$O/${TAG}_$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n ${TAILN:-4} "$O/${TAG}_$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
if [ -n "$KEXPR" ]; then step pytest 1100 python3 -m pytest tests -m gpu -q -x --timeout 600 -k "$KEXPR" || exit 1
else step pytest 1100 python3 -m pytest tests -m gpu -q -x --timeout 600 || exit 1; fi
step bench 500 python3 bench.py --check
step bench_r8 300 python3 bench.py --steps 40 --warmup 5 --cpu-col-stride 0 --emulate-rank-of 8
if [ -n "$TRACE" ]; then
  step trace_r8 300 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_trace_r8 -- python3 bench.py --steps 12 --warmup 3 --cpu-col-stride 0 --emulate-rank-of 8 --same-camera-steps 12
  find $O/${TAG}_trace_r8 -name "*kernel_trace.csv" -exec cp {} $O/${TAG}_trace_r8.csv \;
  rm -rf $O/${TAG}_trace_r8
fi
echo "session done"
