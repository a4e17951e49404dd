#!/usr/bin/env python3
"""A/B harness for kernel tuning on the GPU box: runs bench.py once per environment variant (each in
its own process, interleaved for several rounds) and prints the trace-kernel time per variant.

    python tools/ab.py --rounds 2 "-" "RBRT_Y_LOW=32" "RBRT_SHARE_IDLE=0 RBRT_LEAF_ROUND=8"
"""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+", help='space-separated KEY=VALUE lists, "-" for the default build')
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--isolated-steps", type=int, default=6, help="isolated launches timed per run (more = less noise)")
    ap.add_argument("--extra", default="", help="extra bench.py arguments")
    args = ap.parse_args()
    res = {v: [] for v in args.variants}
    for r in range(args.rounds):
        for v in args.variants:
            env = dict(os.environ, RBRT_HIP_LAB="1")
            if v != "-":
                for kv in v.split():
                    k, val = kv.split("=", 1)
                    env[k] = val
            cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", str(args.steps), "--warmup", "2",
                   "--cpu-col-stride", "0", "--single-frames", "0", "--isolated-steps", str(args.isolated_steps)] + args.extra.split()
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            if p.returncode != 0:
                print(f"[{v}] FAILED rc={p.returncode}: {p.stderr[-400:]}", flush=True)
                continue
            j = json.loads(p.stdout.strip().splitlines()[-1])
            res[v].append((j["roofline"]["kernel_ms"], j["value"], j["roofline"]["frac"], j["ms_per_step"]))
            c = j["roofline"]["counters"]
            print(f"round {r} [{v}] isolated_kernel_ms={j['roofline']['kernel_ms']} ms_per_step={j['ms_per_step']} value={j['value']} "
                  f"nodes={c['nodes_visited'] / 1e6:.1f}M tris={c['tris_tested'] / 1e6:.1f}M sha={j['config']['image_sha256_16']}", flush=True)
    print("---- summary (min / median kernel_ms) ----")
    for v, xs in res.items():
        if xs:
            ks = sorted(x[0] for x in xs)
            print(f"{v:50s} min {ks[0]:.3f}  med {ks[len(ks)//2]:.3f}  value(max) {max(x[1] for x in xs):.1f}  ms/step(min) {min(x[3] for x in xs):.3f}")


if __name__ == "__main__":
    main()
