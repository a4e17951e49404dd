#!/usr/bin/env python3
"""SAH cost of the 4-wide trees the three builders make for the same mesh (host binned SAH, GPU PLOC, GPU LBVH).

    cost = sum over inner nodes area(node)/area(root) * C_node + sum over leaves area(leaf)/area(root) * n_tris * C_tri

with the child boxes as stored in the nodes (C_node = 1 per 4-wide visit, C_tri = 0.35: the kernel's instruction
ratio). A script, not a test (it uses the oracle's mesh preparation, so it lives under tests/). Run on the GPU box:
python tests/bvh_quality.py [n_triangles ...]"""
import os
os.environ.setdefault("RBRT_HIP_LAB", "1")  # the knobs below are lab knobs (include/rbrt_hip_debug.h)
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import scenes  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
from test_bvh_device import build_device  # noqa: E402
from test_bvh_host import build as build_host, LEAF_BITS, NO_CHILD  # noqa: E402


def cost(N, c_tri=0.35):
    lo = N[:, 0:12].reshape(-1, 3, 4)
    hi = N[:, 12:24].reshape(-1, 3, 4)
    link = N[:, 24:28].view(np.int32)
    ext = np.maximum(hi - lo, 0.0)
    area = 2.0 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0])  # [n, 4] child areas
    used = link != NO_CHILD
    area = np.where(used, area, 0.0)  # (unused slots hold the empty box: lo = +inf, hi = -inf)
    root_lo = np.where(used[0], lo[0], np.inf).min(1)
    root_hi = np.where(used[0], hi[0], -np.inf).max(1)
    e = root_hi - root_lo
    root_area = 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0])
    inner = used & (link >= 0)
    leaf = used & (link < 0)
    n_tris = ((~link) & ((1 << LEAF_BITS) - 1)) + 1
    c_inner = 1.0 + float((area * inner).sum() / root_area)          # the root visit + every inner child's
    c_leaf = float((area * leaf * n_tris).sum() / root_area) * c_tri
    return c_inner, c_leaf, int(inner.sum()) + 1, int(leaf.sum()), float((n_tris * leaf).sum() / max(1, leaf.sum()))


def cost_by_depth(N):
    """Inner-node part of the cost per depth of the 4-wide tree (root = 0)."""
    lo = N[:, 0:12].reshape(-1, 3, 4)
    hi = N[:, 12:24].reshape(-1, 3, 4)
    link = N[:, 24:28].view(np.int32)
    ext = np.maximum(hi - lo, 0.0)
    area = 2.0 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0])
    used = link != NO_CHILD
    area = np.where(used, area, 0.0)
    root_lo = np.where(used[0], lo[0], np.inf).min(1)
    root_hi = np.where(used[0], hi[0], -np.inf).max(1)
    e = root_hi - root_lo
    root_area = 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0])
    out = []
    level = np.array([0])
    while len(level):
        inner = used[level] & (link[level] >= 0)
        out.append(float((area[level] * inner).sum() / root_area))
        level = link[level][inner]
    return out


def main():
    oracle.lib()
    for n in [int(a) for a in sys.argv[1:]] or [69451, 871414]:
        md = scenes.standin_mesh(oracle, n, **scenes.EXAMPLE_MESH)
        rows = [("host SAH", build_host(md))]
        for algo in ("ploc", "lbvh"):
            os.environ["RBRT_BVH_DEVICE_ALGO"] = algo
            rows.append((f"gpu {algo}", build_device(md)))
        for r in os.environ.get("RBRT_QUALITY_RADII", "").split():
            os.environ["RBRT_BVH_DEVICE_ALGO"] = "ploc"
            os.environ["RBRT_PLOC_RADIUS"] = r
            rows.append((f"gpu ploc r={r}", build_device(md)))
            os.environ.pop("RBRT_PLOC_RADIUS")
        print(f"{n} triangles")
        for name, (N, T, depth, _) in rows:
            ci, cl, ni, nl, tpl = cost(N)
            print(f"  {name:16s} cost {ci + cl:8.2f} (nodes {ci:7.2f} + leaves {cl:6.2f})  inner {ni:7d} leaves {nl:7d} "
                  f"tris/leaf {tpl:.2f} depth {depth}")
            if os.environ.get("RBRT_QUALITY_LEVELS"):
                print("      by depth:", " ".join(f"{c:.2f}" for c in cost_by_depth(N)))


if __name__ == "__main__":
    main()
