#!/bin/bash
# tools/bench_repeat.sh TAG N [ENV=VALUE ...]: N default bench runs (no CPU leg), one line each: step, isolated launch, launch mix.
# BENCH_ARGS: extra bench.py arguments (default: --steps 20 --warmup 3)
TAG=$1; N=$2; shift 2
mkdir -p gpurun_out/r3
for i in $(seq 1 "$N"); do
  env RBRT_HIP_LAB=1 "$@" timeout -k 10 200 python3 bench.py ${BENCH_ARGS:---steps 20 --warmup 3} --cpu-col-stride 0 --single-frames 0 > gpurun_out/r3/rep_${TAG}_$i.json 2> gpurun_out/r3/rep_${TAG}_$i.err || { echo "run $i failed"; tail -3 gpurun_out/r3/rep_${TAG}_$i.err; exit 1; }
  python3 - "$TAG" "$i" <<'PY'
import json, sys
tag, i = sys.argv[1], sys.argv[2]
j = json.loads(open(f"gpurun_out/r3/rep_{tag}_{i}.json").read().strip().splitlines()[-1])
print(tag, i, "ms_per_step", j["ms_per_step"], "isolated", j["roofline"]["kernel_ms"], "mix", j["config"]["launch_mix_timed_region"], "host_issue", j["config"]["host_issue_ms_per_step"], "value", j["value"], flush=True)
PY
done
