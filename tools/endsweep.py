#!/usr/bin/env python3
"""A/B of scheduling knobs in ONE process (run on the GPU box): every variant is a set of lab knobs (read by
rbrt_hip_scene_create), so a variant is a fresh HipScene of the same host scene. Per variant and round:

  frame   pipelined ms per step / isolated launch ms (HIP events) / one blocking frame with the RGB8 copy
  eighth  rank 0's share of an 8-GPU run: pipelined ms per step / isolated launch ms

    python3 tools/endsweep.py --rounds 3 "-" "RBRT_Y_LOW=32" "RBRT_SHARE_IDLE=8 PIPE=4"

Pseudo-knobs handled here: PIPE=n (rbrt_hip_scene_set_pipeline(n) for the pipelined leg), DEPTH=n (opts.max_depth = n:
diagnosis only, it changes the image), STATS=1 (print the counting build's end-of-launch counters once).
The image SHA of every variant is printed (it must not change).
"""
import argparse
import hashlib
import os
import statistics
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--iso", type=int, default=6)
    ap.add_argument("--triangles", type=int, default=69451)
    ap.add_argument("--mesh", default="smooth")
    ap.add_argument("--mesh-scale", type=float, default=None)
    ap.add_argument("--mesh-translation", default=None, help="x,y,z")
    ap.add_argument("--scene", default=str(ROOT / "scenes" / "example_scene.yaml"))
    ap.add_argument("--new-camera", type=int, default=0, help="1: every frame of the pipelined leg has a camera the library has not seen")
    ap.add_argument("--worlds", default="1,8", help="tile_world values to time (rank 0's share)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--spp", type=int, default=50)
    args = ap.parse_args()
    os.environ["RBRT_HIP_LAB"] = "1"
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (as bench.py and the CLI do: the library's pipeline is eight streams deep)
    import torch

    import rbrt_amd
    from rbrt_amd import abi, standin
    work = Path(tempfile.mkdtemp(prefix="rbrt_sweep_"))
    obj = standin.ensure_obj(work / "bunny.obj", args.triangles, args.mesh)
    text = Path(args.scene).read_text().replace("obj_filepath: bunny.obj", f"obj_filepath: {obj}")
    if args.mesh_scale is not None or args.mesh_translation:
        import yaml
        doc = yaml.safe_load(text)
        for mb in doc.get("mesh_blueprints") or []:
            if args.mesh_scale is not None:
                mb["scale"] = float(args.mesh_scale)
            if args.mesh_translation:
                mb["translation"] = dict(zip("xyz", (float(v) for v in args.mesh_translation.split(","))))
        text = "---\n" + yaml.safe_dump(doc, sort_keys=False)
    (work / "scene.yaml").write_text(text)
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    os.dup2(devnull, 1)
    try:
        hs = abi.HostScene(work / "scene.yaml", args.height, args.width)
    finally:
        os.dup2(saved, 1)
    cam = hs.camera
    W, H = args.width, args.height
    import numpy as np
    cam_no = [0]

    def next_cam():
        if not args.new_camera:
            return cam
        cam_no[0] += 1
        c = type(cam).from_buffer_copy(cam)
        c.position[0] = float(np.float32(cam.position[0]) + np.float32(cam_no[0] % 4096) * np.spacing(np.float32(cam.position[0])))
        return c

    stream = torch.cuda.current_stream().cuda_stream
    img = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    host_rgb = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
    worlds = [int(x) for x in args.worlds.split(",")]
    res = {v: {w: {"step": [], "iso": [], "single": []} for w in worlds} for v in args.variants}
    shas = {}
    mixes = {}
    knob_names = set()
    for v in args.variants:
        if v != "-":
            knob_names.update(kv.split("=", 1)[0] for kv in v.split() if kv.split("=", 1)[0] not in ("PIPE", "DEPTH", "STATS"))
    for r in range(args.rounds):
        for v in args.variants:
            for k in knob_names:
                os.environ.pop(k, None)
            pseudo = {}
            if v != "-":
                for kv in v.split():
                    k, val = kv.split("=", 1)
                    if k in ("PIPE", "DEPTH", "STATS"):
                        pseudo[k] = int(val)
                    else:
                        os.environ[k] = val
            scene = rbrt_amd.HipScene(hs)
            scene.refine_wait(300.0)  # (measured on the tree a handle goes on with: api.cpp struct Refine)
            for w in worlds:
                opts = abi.default_opts(spp=args.spp, seed=1, tile_rank=0, tile_world=w)
                if "DEPTH" in pseudo:
                    opts.max_depth = pseudo["DEPTH"]
                if pseudo.get("STATS") and r == 0:
                    so = abi.default_opts(spp=args.spp, seed=1, tile_rank=0, tile_world=w, flags=abi.FLAG_COLLECT_STATS)
                    scene.render_device(cam, so, img.data_ptr(), None, stream)
                    torch.cuda.synchronize()
                    d = scene.debug_counters()
                    print(f"stats [{v}] w{w}:", {k: d[k] for k in ("passes_term", "passes_lamb", "passes_metal", "passes_diel", "slots_term", "slots_lamb", "drain_slowest", "drain_sum", "shared_entries_given", "path_len_hist")}, flush=True)
                scene.set_pipeline(pseudo.get("PIPE", 0))
                for _ in range(3):
                    scene.render_device(next_cam(), opts, img.data_ptr(), None, stream)
                torch.cuda.synchronize()
                scene.set_timing(True)  # (resets the launch mix: how many launches of the timed loop took the full grid / half of it)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    scene.render_device(next_cam(), opts, img.data_ptr(), None, stream)
                torch.cuda.synchronize()
                res[v][w]["step"].append((time.perf_counter() - t0) / args.steps * 1e3)
                mixes.setdefault((v, w), []).append(scene.launch_mix())
                scene.set_timing(False)
                if args.new_camera:  # (the frame whose hash is printed: the scene's own camera)
                    scene.render_device(cam, opts, img.data_ptr(), None, stream)
                    torch.cuda.synchronize()
                if w == 1 and v not in shas:
                    shas[v] = hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:16]
                # blocking frames with the RGB8 copy (what render_scene returns)
                ts = []
                scene.render_device(cam, opts, None, rgb8.data_ptr(), stream)
                torch.cuda.synchronize()
                for _ in range(4):
                    t0 = time.perf_counter()
                    scene.render_device(cam, opts, None, rgb8.data_ptr(), stream)
                    host_rgb.copy_(rgb8, non_blocking=True)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                res[v][w]["single"].append(sum(ts) / len(ts))
                scene.set_pipeline(1)
                scene.render_device(cam, opts, img.data_ptr(), None, stream)
                torch.cuda.synchronize()
                scene.set_timing(True)
                for _ in range(args.iso):
                    scene.render_device(cam, opts, img.data_ptr(), None, stream)
                torch.cuda.synchronize()
                tr, _, n = scene.kernel_ms()
                scene.set_timing(False)
                res[v][w]["iso"].append(tr / max(1, n))
            scene.check()
            scene.close()
            line = "  ".join(f"w{w}: step {res[v][w]['step'][-1]:.3f} iso {res[v][w]['iso'][-1]:.3f} single {res[v][w]['single'][-1]:.3f}" for w in worlds)
            print(f"round {r} [{v}] {line} sha {shas.get(v)}", flush=True)
    print("---- summary: median (min) ms ----")
    for v in args.variants:
        line = "  ".join(f"w{w}: step {statistics.median(res[v][w]['step']):.3f} ({min(res[v][w]['step']):.3f}) iso {statistics.median(res[v][w]['iso']):.3f} "
                         f"({min(res[v][w]['iso']):.3f}) single {statistics.median(res[v][w]['single']):.3f}" for w in worlds)
        print(f"{v:60s} {line}  sha {shas.get(v)}  full/half grids {[mixes[(v, w)][-1] for w in worlds]}")


if __name__ == "__main__":
    main()
