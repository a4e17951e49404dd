#!/bin/bash
OUT=gpurun_out/r5/s16; mkdir -p $OUT
timeout -k 10 900 python3 tools/endsweep.py --rounds 8 --new-camera 1 --worlds 2,4 --iso 1 "-" "RBRT_HELPERS=0" > $OUT/helpers_w24.log 2>&1
python3 - <<'PY'
import re, collections, statistics
d=collections.defaultdict(list)
for l in open('gpurun_out/r5/s16/helpers_w24.log'):
    for m in re.finditer(r'round \d+ \[(.*?)\] (.*)', l):
        for w,st in re.findall(r'w(\d+): step ([\d.]+)', m.group(2)): d[(m.group(1), w)].append(float(st))
for k,v in d.items(): print(f"{k[0]:20s} w{k[1]} median {statistics.median(v):.3f} min {min(v):.3f} max {max(v):.3f}")
PY
