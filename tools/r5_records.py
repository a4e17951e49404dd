#!/usr/bin/env python3
"""Turns the logs of tools/r5_final.sh (gpurun_out/r5/final/) into the round's records under profiles/:
r05_bench_driver_form.json, r05_bench_configs.txt, r05_rank_shares.json (its `records`), r05_create_sweep.txt, r05_cli_report.json."""
import json
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
F = ROOT / "gpurun_out" / "r5" / "final"
P = ROOT / "profiles"


def last(name):
    lines = [l for l in (F / f"{name}.log").read_text().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


drv = [last(f"driver_{i}") for i in (1, 2, 3)]
(P / "r05_bench_driver_form.json").write_text(json.dumps(
    {"what": "the driver's command three times (python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-col-stride 0) on one box, the final build of round 5",
     "lines": drv}) + "\n")
rows = ["# bench.py --steps 100 --warmup 5 --config X (tools/r5_final.sh bench): Mray-samples/s, ms per step, isolated launch ms, roofline frac, "
        "nodes / tris per sample, blocking frame ms, one_shot ms, image sha"]
for c in ("2", "2r", "4", "4v"):
    j = last(f"cfg_{c}")
    r = j["roofline"]
    rows.append(f"{c:3s} {j['value']:9.1f} {j['ms_per_step']:7.4f} {r['kernel_ms']:7.4f} {r['frac']:6.4f} {r['per_sample']['nodes_visited']:6.3f} {r['per_sample']['tris_tested']:6.3f} "
                f"{j['single_frame']['ms']:7.3f} {j['one_shot']['ms']:7.3f} {j['config']['image_sha256_16']}")
(P / "r05_bench_configs.txt").write_text("\n".join(rows) + "\n")
sh = json.loads((P / "r05_rank_shares.json").read_text())
sh["records"] = {"config2_1024x768x50": last("shares2"), "config3_1920x1080x512": last("shares3")}
(P / "r05_rank_shares.json").write_text(json.dumps(sh, indent=1) + "\n")
sweep = (F / "sweep.log").read_text()
(P / "r05_create_sweep.txt").write_text(sweep)
shutil.copy(F / "cli" / "rep_b.json", P / "r05_cli_report.json")
print("\n".join(rows))
print("driver:", [(d["ms_per_step"], d["value"], d["one_shot"]["ms"]) for d in drv])
