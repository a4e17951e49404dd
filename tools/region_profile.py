#!/usr/bin/env python3
"""Where a wave's lifetime goes, region by region (analysis aid; run on the GPU box).

    make timers && RBRT_HIP_LIB=rbrt_amd/lib/librbrt_hip_timers.so python3 tools/region_profile.py [--triangles N] [--frames K]

The timers build of the library (-DRBRT_REGION_TIMERS=1, megakernel.inl) stamps s_memtime at every region marker and
adds the cycles since the previous marker to the region that just ended. This prints, for BASELINE config 2, each
region's share of the summed wave lifetimes -- issue time AND waits (s_waitcnt, instruction fetch, arbitration between
the four waves of a SIMD), which instruction counts cannot show. The stamps cost about 3 % (an SMEM read + a wait each).
"""
import argparse
import os
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
REGIONS = ("init finalise census refill choose burst_top leaf_round leaf_chunk walk burst_end pass_lists term gen scatter_load "
           "round_top scatter_kind spheres_gate gate park").split()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--triangles", type=int, default=69451)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--spp", type=int, default=50)
    ap.add_argument("--emulate-rank-of", type=int, default=0)
    args = ap.parse_args()
    import torch

    import rbrt_amd
    from rbrt_amd import abi, standin
    if "timers" not in os.environ.get("RBRT_HIP_LIB", ""):
        raise SystemExit("set RBRT_HIP_LIB to the timers build (make timers)")
    work = Path(tempfile.mkdtemp(prefix="rbrt_regions_"))
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
    scene.set_pipeline(1)
    print("# scene:", scene.info())
    world = args.emulate_rank_of or 1
    opts = abi.default_opts(spp=args.spp, seed=1, tile_rank=0, tile_world=world)
    img = torch.empty((args.height, args.width, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_device(hs.camera, opts, img.data_ptr(), None, stream)  # warm-up
    torch.cuda.synchronize()
    nr = len(REGIONS)
    before = scene.raw_debug_counters()
    scene.set_timing(True)
    phases, slowest = [], []
    for _ in range(args.frames):
        for k in (1, 3):
            scene.set_debug_counter(nr + k, 2**64 - 1)   # minima
        scene.set_debug_counter(nr + 8, scene.raw_debug_counters()[nr + 5])  # epoch of the start histogram: the previous launch's end
        for k in (2, 4, 5, 9, 10):
            scene.set_debug_counter(nr + k, 0)            # maxima
        scene.render_device(hs.camera, opts, img.data_ptr(), None, stream)
        torch.cuda.synchronize()
        c = scene.raw_debug_counters()
        first_start, last_start, first_out, last_out, last_end = (c[nr + k] for k in (1, 2, 3, 4, 5))
        slow, slow_b = c[nr + 9], c[nr + 10]
        slowest.append(dict(drain_us=(slow >> 40) / 100.0, rounds=(slow >> 30) & 1023, passes=(slow >> 20) & 1023, trav_steps=(slow >> 8) & 4095,
                            paths_in_hand=slow & 255, longest_path_bounces=slow_b & 0xFFFF))
        phases.append(((last_start - first_start) / 100.0, (first_out - first_start) / 100.0, (last_out - first_start) / 100.0,
                       (last_end - first_start) / 100.0))
    ms, _, n = scene.kernel_ms()
    after = scene.raw_debug_counters()
    d = [a - b for a, b in zip(after, before)]
    total, waves = sum(d[:len(REGIONS)]), d[len(REGIONS)]
    print(f"# {args.width}x{args.height}x{args.spp}, {args.triangles} triangles, {n} launches, {ms / n:.3f} ms per launch (with stamps), "
          f"{waves // max(1, n)} waves, {total / max(1, waves) / 1e6:.3f} M cycles per wave")
    groups = {"scheduling": ("finalise", "census", "refill", "choose", "pass_lists"),
              "traversal": ("burst_top", "walk", "burst_end"), "leaf rounds": ("leaf_round", "leaf_chunk"),
              "terminate + generate": ("term", "gen"), "scatter": ("scatter_load", "round_top", "scatter_kind"),
              "spheres + gate": ("spheres_gate", "gate"), "park": ("park",), "init": ("init",)}
    for name, c in zip(REGIONS, d):
        print(f"{name:14s} {c / total * 100:6.2f} %   {c / max(1, waves) / 1e3:10.1f} k cycles per wave")
    waves_per = waves // max(1, n)
    print(f"# wall clock of a launch, us from the first wave's start (mean of {len(phases)} launches): last wave started "
          f"{sum(p[0] for p in phases) / len(phases):.1f}, first wave out of work {sum(p[1] for p in phases) / len(phases):.1f}, last wave out of work "
          f"{sum(p[2] for p in phases) / len(phases):.1f}, last wave ended {sum(p[3] for p in phases) / len(phases):.1f}")
    print(f"# summed drain time of the waves / summed lifetime: {d[nr + 6] / max(1, d[nr + 7]) * 100:.1f} %  ({d[nr + 6] / max(1, waves) / 100:.1f} us of "
          f"{d[nr + 7] / max(1, waves) / 100:.1f} us per wave)")
    print("# the wave with the longest drain, per launch:", slowest)
    hist = [x // max(1, n) for x in d[32:64]]
    print("# waves by start time, 100-us buckets after the previous launch's end (per launch):", hist)
    print("# grouped")
    for g, names in groups.items():
        print(f"{g:22s} {sum(d[REGIONS.index(x)] for x in names) / total * 100:6.2f} %")


if __name__ == "__main__":
    main()
