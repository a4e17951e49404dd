#!/bin/bash
export RBRT_HIP_LAB=1  # the scheduling knobs below are lab knobs (include/rbrt_hip_debug.h)
# SQ / TCP counters of the trace megakernel for one library build: tools/pmc_ab.sh <tag> [ENV=VALUE ...]
# (two rocprofv3 --pmc passes over a short unpipelined bench run; prints per-dispatch averages)
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
B="python3 bench.py --steps 3 --warmup 1 --pipeline 1 --cpu-col-stride 0 --single-frames 0"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS \
    --output-format csv -d "$OUT/sq" -- $B > "$OUT/b1.json" 2> "$OUT/b1.err" || tail -3 "$OUT/b1.err"
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS \
    --output-format csv -d "$OUT/tcp" -- $B > "$OUT/b2.json" 2> "$OUT/b2.err" || tail -3 "$OUT/b2.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
per = collections.defaultdict(float); nd = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trace_megakernel" in r["Kernel_Name"] and re.search(r"<\d+, false, (true|false), false>", r["Kernel_Name"]):
            per[r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(per): print(f"{k:32s} {per[k] / max(1, len(nd[k])):.4g}")
if "SQ_ACTIVE_INST_VALU" in per: print("valu_lane_utilisation", per["SQ_THREAD_CYCLES_VALU"] / (64 * per["SQ_ACTIVE_INST_VALU"]))
PY
