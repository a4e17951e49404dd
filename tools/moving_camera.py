#!/usr/bin/env python3
"""Throughput of a stream of frames whose camera moves between frames against the same stream with a fixed camera (GPU
box): with a new camera every frame, every launch pays the tile pass (primary_cull_kernel + tile_lists_kernel) on its
lane's stream instead of reusing the lane's lists.

    python3 tools/moving_camera.py [--frames 40]
"""
import argparse
import copy
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--triangles", type=int, default=69451)
    args = ap.parse_args()
    import torch

    import rbrt_amd
    from rbrt_amd import abi, standin
    work = Path(tempfile.mkdtemp(prefix="rbrt_move_"))
    obj = standin.ensure_obj(work / "bunny.obj", args.triangles)
    (work / "scene.yaml").write_text((ROOT / "scenes" / "example_scene.yaml").read_text().replace("obj_filepath: bunny.obj", f"obj_filepath: {obj}"))
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    os.dup2(devnull, 1)
    try:
        hs = abi.HostScene(work / "scene.yaml", 768, 1024)
    finally:
        os.dup2(saved, 1)
    scene = rbrt_amd.HipScene(hs)
    scene.refine_wait(300.0)  # (measured on the tree a handle goes on with: api.cpp struct Refine)
    opts = abi.default_opts(spp=50, seed=1)
    img = torch.empty((768, 1024, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def cam_at(k):
        c = copy.copy(hs.camera)  # (a ctypes structure: a shallow copy is a new struct)
        c = type(hs.camera).from_buffer_copy(hs.camera)
        dx = 0.002 * k  # a sideways drift: position and image centre move together
        c.position[0] += dx
        c.img_center_point[0] += dx
        return c

    for label, moving in (("fixed", False), ("moving", True), ("fixed", False), ("moving", True)):
        for k in range(4):
            scene.render_device(cam_at(k if moving else 0), opts, img.data_ptr(), None, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.frames):
            scene.render_device(cam_at(k + 4 if moving else 0), opts, img.data_ptr(), None, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.frames
        print(f"{label:7s} camera: {dt * 1e3:.3f} ms per frame", flush=True)


if __name__ == "__main__":
    main()
