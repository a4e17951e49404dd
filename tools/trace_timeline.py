#!/usr/bin/env python3
"""Prints the timeline of a rocprofv3 --kernel-trace CSV for this library's kernels (start, end, duration, queue), from the
first trace launch on: tools/trace_timeline.py FILE.csv [max_rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 10**9
names = (("trace_megakernel", "trace"), ("sky_resolve", "sky_resolve"), ("resolve_kernel", "resolve"), ("primary_cull", "primary_cull"), ("tile_lists", "tile_lists"))
out = []
for r in rows:
    short = next((s for key, s in names if key in r["Kernel_Name"]), None)
    if short:
        out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r["Queue_Id"]), int(r["Grid_Size_X"])))
out.sort()
first = next(o for o in out if o[2] == "trace")[0]
prev_start = None
for s, e, k, q, g in out[:limit]:
    gap = "" if k != "trace" or prev_start is None else f"  (+{(s - prev_start) / 1e3:7.1f} after the previous trace launch's start)"
    if k == "trace":
        prev_start = s
    print(f"{(s - first) / 1e3:10.1f} {(e - first) / 1e3:10.1f}  dur {(e - s) / 1e3:8.1f} us  q{q}  {k:12s} threads {g}{gap}")
