#!/bin/bash
# Instruction-cache counters of one isolated trace launch (run on the GPU box through gpurun): is the 36-KB kernel served from the
# 64-KB instruction cache two CUs share?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_icache; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/p -- python3 bench.py --steps 2 --warmup 0 --pipeline 1 --cpu-col-stride 0 --single-frames 0 --one-shot 0 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_icache/p/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "trace_megakernel" in k or "resolve" in k:
        print(k, {a: round(b) for a, b in v.items()})
PY
