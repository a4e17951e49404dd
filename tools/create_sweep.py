#!/usr/bin/env python3
"""What rbrt_hip_scene_create costs with each BVH builder, over mesh sizes: the measurement behind api.cpp's
device_builder_is_cheaper (kHostBuildSecPerTri, kDeviceBuildSec0, kDeviceBuildSecPerTri).

    python tools/create_sweep.py [--sizes 500,2000,...] [--reps 3] > profiles/r05_create_sweep.txt

Per size and builder: the fastest of `reps` creations, split as rbrt_hip_scene_create_times reports it (ms); and, for the
builder the cost rule picks, how long the background thread took to put the host builder's tree on the device."""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="500,1000,2000,4000,8000,16000,32000,69451,131072,262144,871414")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import rbrt_amd
    import scenes
    from oracle import pyoracle
    print(f"# host threads: {len(os.sched_getaffinity(0))}; ms; fastest of {args.reps}")
    print(f"{'entries':>8} {'builder':>7} {'create':>8} {'hip_init':>8} {'upload':>8} {'build':>8} {'lanes':>8} {'nodes':>8}")
    for n in (int(x) for x in args.sizes.split(",")):
        sc = scenes.example_scene(pyoracle, n)
        for builder in ("host", "device"):
            os.environ["RBRT_BVH_BUILDER"] = builder
            best = None
            for _ in range(args.reps):
                with rbrt_amd.HipScene(sc) as hs:
                    t = hs.create_times()
                    t["nodes"] = hs.info()["n_nodes"]
                if best is None or t["create_s"] < best["create_s"]:
                    best = t
            print(f"{n:>8} {builder:>7} {best['create_s'] * 1e3:8.2f} {best['hip_init_s'] * 1e3:8.2f} {best['upload_s'] * 1e3:8.2f} "
                  f"{best['bvh_build_s'] * 1e3:8.2f} {best['lanes_s'] * 1e3:8.2f} {best['nodes']:>8}", flush=True)
        os.environ.pop("RBRT_BVH_BUILDER")
        with rbrt_amd.HipScene(sc) as hs:
            t = hs.create_times()
            t0 = time.perf_counter()
            state, secs = hs.refine_wait(300.0)
            print(f"{n:>8} {'rule':>7} {t['create_s'] * 1e3:8.2f} first tree by the {'device' if t['meshes_device_built'] else 'host'} builder; "
                  f"background tree: state {state}, {secs * 1e3:.1f} ms (waited {1e3 * (time.perf_counter() - t0):.1f} ms)", flush=True)


if __name__ == "__main__":
    main()
