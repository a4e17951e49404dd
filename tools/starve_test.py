#!/usr/bin/env python3
"""Does a short kernel with big workgroups and LDS -- the shape of a collective's copy kernel -- get onto the GPU
between the persistent trace launches? (The 8-GPU gather cannot be run on the one-GPU box; this puts a stand-in on
the caller's stream after every frame of rank 0's eighth and looks at the step time.)

    python tools/starve_test.py            # on the GPU box
"""
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import rbrt_amd  # noqa: E402
from rbrt_amd import abi, standin  # noqa: E402


def main():
    torch.cuda.set_device(0)
    work = Path(tempfile.mkdtemp(prefix="rbrt_starve_"))
    obj = standin.ensure_obj(work / "bunny.obj", 69451)
    (work / "scene.yaml").write_text((ROOT / "scenes" / "example_scene.yaml").read_text().replace("obj_filepath: bunny.obj", f"obj_filepath: {obj}"))
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    os.dup2(devnull, 1)
    try:
        host_scene = abi.HostScene(work / "scene.yaml", 768, 1024)
    finally:
        os.dup2(saved, 1)
    scene = rbrt_amd.HipScene(host_scene, device=0)
    scene.refine_wait(300.0)  # (measured on the tree a handle goes on with: api.cpp struct Refine)
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.randn(512, 512, device="cuda")
    b = torch.randn(512, 512, device="cuda")
    big = torch.randn(1 << 22, device="cuda")
    for world in (1, 8):
        opts = abi.default_opts(spp=50, seed=1, tile_rank=0, tile_world=world)
        n = rbrt_amd.packed_pixels(1024, 768, 0, world) * 3 if world > 1 else 1024 * 768 * 3
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        steps = 20 if world == 1 else 80
        for kind in ("none", "gemm 512^3 (LDS-tiled, 256-thread workgroups)", "cumsum 4M (multi-pass scan)", "copy 16 MB"):
            def extra():
                if kind.startswith("gemm"):
                    torch.mm(a, b)
                elif kind.startswith("cumsum"):
                    torch.cumsum(big, 0)
                elif kind.startswith("copy"):
                    big.clone()
            for _ in range(4):
                scene.render_device(host_scene.camera, opts, out.data_ptr(), None, stream)
                extra()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                scene.render_device(host_scene.camera, opts, out.data_ptr(), None, stream)
                extra()
            torch.cuda.synchronize()
            print(f"1/{world} of the frame, after every frame: {kind:50s} {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
