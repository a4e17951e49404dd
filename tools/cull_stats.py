#!/usr/bin/env python3
"""What the primary-ray culling table says for the bench workload (GPU box): per object, the share of tiles whose
camera rays cannot reach it; the share of background-only tiles; and the pass / lane counts of a counting render with
the culling on and off (RBRT_PRIMARY_CULL is a lab knob: the script sets RBRT_HIP_LAB=1 for its children).

    python3 tools/cull_stats.py [--width 1024 --height 768 --triangles 69451]
"""
import argparse
import os
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--triangles", type=int, default=69451)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=768)
    args = ap.parse_args()
    import numpy as np

    import rbrt_amd
    from rbrt_amd import abi, standin
    work = Path(tempfile.mkdtemp(prefix="rbrt_cull_"))
    obj = standin.ensure_obj(work / "bunny.obj", args.triangles)
    (work / "scene.yaml").write_text((ROOT / "scenes" / "example_scene.yaml").read_text().replace("obj_filepath: bunny.obj", f"obj_filepath: {obj}"))
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    os.dup2(devnull, 1)
    try:
        hs = abi.HostScene(work / "scene.yaml", args.height, args.width)
    finally:
        os.dup2(saved, 1)
    scene = rbrt_amd.HipScene(hs)
    scene.refine_wait(300.0)  # (measured on the tree a handle goes on with: api.cpp struct Refine)
    info = scene.info()
    t = scene.primary_cull(hs.camera)
    n = t.size
    print(f"# {args.width}x{args.height}: {t.shape[1]} x {t.shape[0]} tiles")
    for e in range(info["n_spheres"]):
        print(f"element {e}: culled in {100.0 * np.count_nonzero((t >> e) & 1) / n:5.1f} % of the tiles")
    for m in range(info["n_meshes"]):
        print(f"mesh {m} box: culled in {100.0 * np.count_nonzero((t >> (24 + m)) & 1) / n:5.1f} % of the tiles")
    print(f"background only: {100.0 * np.count_nonzero(t >> 31) / n:5.1f} % of the tiles")
    tests = sum(np.count_nonzero(((t >> e) & 1) == 0) for e in range(info["n_spheres"]))
    print(f"sphere tests left per camera ray: {tests / n:.2f} of {info['n_spheres']}")
    rows = ["".join("#" if (w >> 31) else str(bin(~w & ((1 << info["n_spheres"]) - 1)).count("1")) for w in r[::2]) for r in t[::2]]
    print("\n".join(rows))


if __name__ == "__main__":
    main()
