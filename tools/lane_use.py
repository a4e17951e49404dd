#!/usr/bin/env python3
"""How full the wave is in each kind of work (run on the GPU box): the headline frame through the counting build of the
kernel, then lanes per traversal step, per leaf round, per shading pass of each kind, per sphere tail.
    python3 tools/lane_use.py [--triangles N] [--mesh smooth|rough] [--world 1]
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
    ap.add_argument("--mesh", default="smooth")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--spp", type=int, default=50)
    args = ap.parse_args()
    os.environ["RBRT_HIP_LAB"] = "1"
    import torch

    import rbrt_amd
    from rbrt_amd import abi, standin
    work = Path(tempfile.mkdtemp(prefix="rbrt_lanes_"))
    obj = standin.ensure_obj(work / "bunny.obj", args.triangles, args.mesh)
    text = (ROOT / "scenes" / "example_scene.yaml").read_text().replace("obj_filepath: bunny.obj", f"obj_filepath: {obj}")
    (work / "scene.yaml").write_text(text)
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    os.dup2(devnull, 1)
    try:
        hs = abi.HostScene(work / "scene.yaml", args.height, args.width)
    finally:
        os.dup2(saved, 1)
    scene = rbrt_amd.HipScene(hs)
    scene.refine_wait(300.0)  # (measured on the tree a handle goes on with: api.cpp struct Refine)
    img = torch.empty((args.height, args.width, 3), dtype=torch.float32, device="cuda")
    so = abi.default_opts(spp=args.spp, seed=1, tile_rank=0, tile_world=args.world, flags=abi.FLAG_COLLECT_STATS)
    scene.render_device(hs.camera, so, img.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d = scene.debug_counters()
    c = scene.stats()
    print("counters:", c)
    ws, ls = d["trav_wave_steps"], d["trav_lane_steps"]
    print(f"traversal steps (wave)      {ws:>12,}   busy lanes per step {ls / max(1, ws):5.1f}")
    print(f"  of them with a node visit {d['walk_rounds']:>12,}   walking lanes per visit round {d['walk_lanes'] / max(1, d['walk_rounds']):5.1f}")
    print(f"leaf rounds                 {d['leaf_rounds']:>12,}   triangles per round {d['leaf_lanes'] / max(1, d['leaf_rounds']):5.1f}")
    for k in ("term", "lamb", "metal", "diel"):
        print(f"passes {k:5s}                {d['passes_' + k]:>12,}   lanes per pass (incl. later rounds) {d['slots_' + k] / max(1, d['passes_' + k]):5.1f}")
    raw = scene.raw_debug_counters()
    n_pass = sum(d['passes_' + k] for k in ("term", "lamb", "metal", "diel"))
    print(f"free lanes per shading pass {raw[23] / max(1, n_pass):5.1f};  the fullest other scatter kind would fill >= 16 of them in {raw[58] >> 32:,} ({(raw[58] & 0xFFFFFFFF) / max(1, raw[58] >> 32):.1f})")
    print(f"extra shading rounds        {d['shade_extra_rounds']:>12,}")
    print(f"traversal bursts            {d['passes_trav']:>12,}   busy lanes at entry {d['slots_empty'] / max(1, d['passes_trav']):5.1f}   steps per burst {ws / max(1, d['passes_trav']):5.1f}")
    print(f"  at entry: rays parked without a lane {d['parked_at_burst_entry'] / max(1, d['passes_trav']):5.1f}   slots of the fullest shading kind {d['fullest_shading_kind_at_burst_entry'] / max(1, d['passes_trav']):5.1f}")
    print(f"traversals {c['mesh_gate_pass']:,}: {d['traversals_ending_at_root'] / max(1, c['mesh_gate_pass']):.3f} of them end at the root (no child box hit); "
          f"node visits per traversal {c['nodes_visited'] / max(1, c['mesh_gate_pass']):.1f}, triangle tests {c['tris_tested'] / max(1, c['mesh_gate_pass']):.1f}, "
          f"accepted hits {c['mesh_hits'] / max(1, c['mesh_gate_pass']):.3f}")
    print(f"refills                     {d['refill_rounds']:>12,}   lanes per refill {d['slots_trav'] / max(1, d['refill_rounds']):5.1f}")
    print(f"scheduling rounds           {d['sched_rounds']:>12,}")
    print(f"sphere tails                {d['sphere_tail_runs']:>12,}   lanes per tail {d['sphere_tail_lanes'] / max(1, d['sphere_tail_runs']):5.1f}")
    print(f"cycles: traversal {d['cycles_trav'] / max(1, d['cycles_total']):.3f}  shading {d['cycles_shade'] / max(1, d['cycles_total']):.3f} of a wave's life")
    print("drain:", d["drain_sum"], d["drain_slowest"])
    scene.close()


if __name__ == "__main__":
    main()
