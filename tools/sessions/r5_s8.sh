#!/bin/bash
# Round 5, session 8: elastic launches carried by idle lanes' streams: tests, then 20-step streams with and without, alternating.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5/s8; rm -rf $OUT; mkdir -p $OUT
step() { local name=$1 tmo=$2; shift 2; echo "== $name"; local t0=$SECONDS
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1; local rc=$?
  echo "== $name rc=$rc ($((SECONDS - t0)) s)"; tail -n 2 "$OUT/$name.log" | cut -c1-200
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi; return $rc; }
step helpers 600 python3 -m pytest tests/test_helpers.py -m gpu -q --timeout 300 -x || exit 1
B="python3 bench.py --warmup 5 --cpu-col-stride 0 --single-frames 0 --one-shot 0 --isolated-steps 0 --same-camera-steps 0"
for pass in 1 2 3 4; do
  for k in 20 40; do
    step on_${k}_$pass 300 $B --steps $k
    RBRT_HIP_LAB=1 RBRT_HELPERS=0 step off_${k}_$pass 300 $B --steps $k
  done
done
RBRT_HIP_LAB=1 RBRT_TRACE_LAUNCHES=1 timeout -k 10 300 $B --steps 20 > $OUT/trace_on.json 2> $OUT/trace_on.err; grep "helper launch" $OUT/trace_on.err | tail -8
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_on -- $B --steps 20 > $OUT/kt_on.json 2> $OUT/kt_on.err
f=$(ls $OUT/kt_on/*/*kernel_trace.csv | head -1); python3 tools/trace_timeline.py $f > $OUT/timeline_on.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5/s8/o*_*.log')):
    j=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], 'ms', j['ms_per_step'], 'helpers', j['config'].get('helper_launches_timed_region'), j['config']['image_sha256_16'])
PY
echo "session done"
