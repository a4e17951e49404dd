#!/usr/bin/env python3
"""Condenses gpurun_out/profile/ (made by tools/profile.sh on the GPU box) into the committed summaries:

    profiles/<tag>_{isolated,default}_kernel_stats.csv   rocprofv3 --kernel-trace --stats per-kernel tables
    profiles/<tag>_pmc.json              counters per dispatch of the trace kernel + derived ratios
    profiles/<tag>_pmc_summary.json      HBM-side bytes per launch, VALU instruction count, lane utilisation:
                                         read by bench.py for roofline.traffic / measured_hbm / valu_issue
"""
import collections
import csv
import glob
import json
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from rbrt_amd.srchash import kernel_source_sha256  # noqa: E402

SRC = ROOT / "gpurun_out" / "profile"
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = ROOT / "profiles"
out_dir.mkdir(exist_ok=True)

if not SRC.is_dir() or not any(SRC.iterdir()):
    sys.exit(f"{SRC} is empty: run tools/profile.sh on the GPU box first (nothing written)")
bench = None
for leg in ("isolated", "default"):
    stats = sorted(glob.glob(str(SRC / f"stats_{leg}" / "*" / "*kernel_stats.csv")), key=lambda f: Path(f).stat().st_mtime)
    if stats:
        shutil.copy(stats[-1], out_dir / f"{tag}_{leg}_kernel_stats.csv")
    jf = SRC / f"bench_stats_{leg}.json"
    if jf.exists() and jf.read_text().strip():
        b = json.loads(jf.read_text().strip().splitlines()[-1])
        (out_dir / f"{tag}_{leg}_bench_under_rocprof.json").write_text(json.dumps(b, indent=1))
        bench = bench or b

per = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
meta = {}
# (gpurun merges a run's files into whatever gpurun_out/profile already holds: only the NEWEST file of each pass counts)
newest = {}
for f in glob.glob(str(SRC / "*" / "*" / "*counter_collection.csv")):
    d = Path(f).parent.parent.name
    if d not in newest or Path(f).stat().st_mtime > Path(newest[d]).stat().st_mtime:
        newest[d] = f
for pass_name, f in newest.items():
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trace_megakernel" not in k or not re.search(r"<\d+, false, (true|false), false>", k):  # (the build without counters)
            continue
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[(k, r["Counter_Name"])].add((pass_name, r["Dispatch_Id"]))  # (a counter collected in two passes: averaged over both)
        meta[k] = {"VGPR": r["VGPR_Count"], "SGPR": r["SGPR_Count"], "LDS_Block_Size": r["LDS_Block_Size"],
                   "Grid_Size": r["Grid_Size"], "Workgroup_Size": r["Workgroup_Size"]}
res = {}
for k, d in per.items():
    c = {name: v / max(1, len(ndisp[(k, name)])) for name, v in d.items()}  # per dispatch
    der = {}
    if "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"]:
        der["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        der["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c:
        der["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        der["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(1.0, c["TCP_TOTAL_CACHE_ACCESSES_sum"])
    res[k] = {"per_dispatch": c, "derived": der, "launch": meta[k]}
(out_dir / f"{tag}_pmc.json").write_text(json.dumps(res, indent=1))

# HBM-side traffic per launch, as MI355X_MICROARCH.md "HBM" prescribes: (FETCH_SIZE + WRITE_SIZE) * 1024 with
# FETCH_SIZE doubled on gfx950 (it tallies 128-B requests at 64 B for 16-B-per-lane loads, which is what
# this kernel issues); both the raw and the corrected figure are kept.
for k, d in res.items():
    c = d["per_dispatch"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        raw = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        corrected = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        cfg = bench["config"]["workload"]
        rec = {"kernel": k, "workload": "1024x768x50", "triangles": 69451, "bench_workload": cfg,
               # (recorded by tools/profile.sh on the GPU box, from the tree the counters were measured on)
               "kernel_source_sha256_16": ((SRC / "kernel_source_sha256.txt").read_text().strip()
                                           if (SRC / "kernel_source_sha256.txt").exists() else kernel_source_sha256()),
               "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"], "hbm_bytes_per_launch_raw": raw,
               "hbm_bytes_per_launch": corrected,
               "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 "
                       "half-count for 16-B-per-lane loads, MI355X_MICROARCH.md HBM section); Infinity-Cache hits "
                       "are included in these fabric-side counters"}
        if "TCC_EA0_RDREQ_sum" in c:
            rec["tcc_ea0_rdreq"] = c["TCC_EA0_RDREQ_sum"]
            rec["tcc_ea0_rdreq_32B"] = c.get("TCC_EA0_RDREQ_32B_sum")
        if "SQ_INSTS_VALU" in c:
            rec["sq_insts_valu"] = c["SQ_INSTS_VALU"]
            rec["sq_insts_salu"] = c.get("SQ_INSTS_SALU")
            rec["sq_lds_bank_conflict"] = c.get("SQ_LDS_BANK_CONFLICT")
            for name in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                         "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
                if name in c:
                    rec[name.lower()] = c[name]
            # the executed VALU instructions by class (the counters name f32 add / mul / fma, transcendental, 32- and 64-bit
            # integer and conversions; compares, selects, min / max, moves and lane operations are the rest)
            if "SQ_INSTS_VALU_FMA_F32" in c:
                mix = {n[len("SQ_INSTS_VALU_"):].lower(): c[n] for n in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32",
                                                                        "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT") if n in c}
                mix["other"] = c["SQ_INSTS_VALU"] - sum(mix.values())
                rec["valu_mix"] = mix
            for name in ("SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS_LOAD", "SQ_INSTS_LDS_STORE", "SQ_INSTS_LDS_ATOMIC", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VSKIPPED"):
                if name in c:
                    rec[name.lower()] = c[name]
            rec["valu_lane_utilisation"] = d["derived"].get("valu_lane_utilisation")
            rec["l2_hit_rate"] = d["derived"].get("l2_hit_rate")
            rec["l1_hit_rate"] = d["derived"].get("l1_hit_rate")
        (out_dir / f"{tag}_pmc_summary.json").write_text(json.dumps(rec, indent=1))
        print("summary", json.dumps(rec))
print(json.dumps({k: v["derived"] for k, v in res.items()}, indent=1))
