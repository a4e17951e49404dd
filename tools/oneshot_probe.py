#!/usr/bin/env python3
"""One-shot calls (rbrt_hip_render) of the smooth and the rough stand-in in turn, one process: per call ms and its parts.
Run on the GPU box: python3 tools/oneshot_probe.py [calls [smooth|rough [frames]]] (a handle of that kind alive beside the calls, rendered that many frames)"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import rbrt_amd
import scenes
from oracle import pyoracle as oracle
oracle.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cam = scenes.camera(oracle, 1024, 768)
sc = {k: scenes.example_scene(oracle, 69451, kind=k) for k in ("smooth", "rough")}
main = sys.argv[2] if len(sys.argv) > 2 else ""
hs = None
if main:  # a handle of this kind stays alive beside the calls (its background tree adopted first)
    hs = rbrt_amd.HipScene(sc[main])
    print("main scene:", main, "background tree", hs.refine_wait(300.0), flush=True)
    if len(sys.argv) > 3:
        import torch
        img = torch.empty((768, 1024, 3), dtype=torch.float32, device="cuda")
        from rbrt_amd import abi
        for _ in range(int(sys.argv[3])): hs.render_device(cam, abi.default_opts(spp=50, seed=1), img.data_ptr())
        torch.cuda.synchronize()
        print("main scene rendered", sys.argv[3], "frames", flush=True)
for k in sc: rbrt_amd.render_scene(cam, 50, sc[k], seed=1, want_radiance=False)
for i in range(n):
    for k in ("smooth", "rough"):
        t0 = time.perf_counter()
        rbrt_amd.render_scene(cam, 50, sc[k], seed=1, want_radiance=False)
        dt = (time.perf_counter() - t0) * 1e3
        t = rbrt_amd.last_render_times()
        print(f"{k:7s} {dt:7.2f} ms  create {t['create_s']*1e3:6.2f} (build {t['bvh_build_s']*1e3:5.2f}) render {t['render_s']*1e3:6.2f} destroy {t['destroy_s']*1e3:5.2f}", flush=True)
