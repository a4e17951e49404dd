"""Deterministic procedural STAND-IN meshes for the assets the reference expects but does not ship.

The reference's scenes load `bunny.obj` (README.md:19-20 downloads it; .gitignore:5 ignores *.obj)
and BASELINE.json also names the Stanford dragon. Neither file exists in this environment and there
is no network, so benchmarks and tests use stand-ins with the SAME TRIANGLE COUNT (69,451 and
871,414) and roughly the same extents. Every report that uses them says "stand-in".

Shape: a cube-sphere (6 faces x n x n quads, vertices welded on the cube lattice) displaced by a
few fixed low-frequency lobes. 12*n^2 >= N triangles are generated and the LAST ones (a patch of
the bottom face) are dropped to hit N exactly, which leaves a hole like the real scan's open base
(an odd triangle count cannot be a closed surface anyway).
"""
from __future__ import annotations

import math
from pathlib import Path

import numpy as np

BUNNY_TRIANGLES = 69451
DRAGON_TRIANGLES = 871414

# Stanford bunny extents (model units, before the scene's `scale: 45`)
_CENTER = np.array([-0.017, 0.110, -0.0015])
_RADII = np.array([0.105, 0.075, 0.082])


def _cube_sphere(n: int, warp=None):
    """Welded cube-sphere: returns (dirs[V,3] unit vectors, faces[12*n*n, 3] vertex indices). `warp` maps the lattice
    coordinate in [-1, 1] to the cube coordinate before projection (default: the equal-area tangent warp)."""
    # face frames: (origin corner, u axis, v axis) on the integer lattice [0,n]^3; -y face last
    frames = [
        ((n, 0, 0), (0, 1, 0), (0, 0, 1)),  # +x
        ((0, 0, 0), (0, 0, 1), (0, 1, 0)),  # -x
        ((0, n, 0), (0, 0, 1), (1, 0, 0)),  # +y
        ((0, 0, n), (1, 0, 0), (0, 1, 0)),  # +z
        ((0, 0, 0), (0, 1, 0), (1, 0, 0)),  # -z
        ((0, 0, 0), (1, 0, 0), (0, 0, 1)),  # -y (bottom: its tail is what gets dropped)
    ]
    ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    keys, quads = [], []
    base = 0
    for (o, u, v) in frames:
        o, u, v = np.array(o), np.array(u), np.array(v)
        lat = o[None, None, :] + ii[..., None] * u + jj[..., None] * v  # (n+1,n+1,3)
        keys.append(lat.reshape(-1, 3))
        idx = base + np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        # outward orientation: (u x v) must point away from the cube centre
        outward = np.dot(np.cross(u, v), (o + (u + v) * n / 2) - n / 2) > 0
        if outward:
            t1, t2 = np.stack([a, b, c], -1), np.stack([a, c, d], -1)
        else:
            t1, t2 = np.stack([a, c, b], -1), np.stack([a, d, c], -1)
        quads.append(np.stack([t1, t2], 2).reshape(-1, 3))
        base += (n + 1) * (n + 1)
    lat = np.concatenate(keys).astype(np.int64)
    faces = np.concatenate(quads)
    key = (lat[:, 0] * (n + 1) + lat[:, 1]) * (n + 1) + lat[:, 2]
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    lat_u = lat[first].astype(np.float64)
    # tangent warp makes the cells nearly equal-area on the sphere
    t = 2.0 * lat_u / n - 1.0
    p = np.tan(t * (math.pi / 4.0)) if warp is None else warp(t)
    dirs = p / np.linalg.norm(p, axis=1, keepdims=True)
    return dirs, inv[faces]


def _radius(d: np.ndarray) -> np.ndarray:
    """Smooth radial displacement: body + head lobe + two ear lobes, fixed constants."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    r = 1.0 + 0.10 * np.sin(3.0 * x + 0.5) * np.cos(2.0 * z - 0.3) + 0.06 * np.sin(5.0 * y + 1.1)
    for (cx, cy, cz, amp, sharp) in ((-0.55, 0.75, 0.15, 0.45, 6.0),   # head
                                     (-0.35, 0.92, 0.18, 0.55, 30.0),  # ear
                                     (-0.15, 0.96, -0.10, 0.50, 30.0)):  # ear
        c = np.array([cx, cy, cz])
        c = c / np.linalg.norm(c)
        r = r + amp * np.exp(sharp * (d @ c - 1.0))
    return r


def make_mesh(n_triangles: int):
    """Returns (vertices float32[V,3], faces int32[n_triangles,3])."""
    n = int(math.ceil(math.sqrt(n_triangles / 12.0)))
    dirs, faces = _cube_sphere(n)
    assert len(faces) >= n_triangles
    faces = faces[:n_triangles]
    pos = _CENTER + dirs * _RADII * _radius(dirs)[:, None] / 1.45
    used = np.unique(faces)
    remap = -np.ones(len(pos), np.int64)
    remap[used] = np.arange(len(used))
    return pos[used].astype(np.float32), remap[faces].astype(np.int32)


def _rough_radius(d: np.ndarray) -> np.ndarray:
    """The ROUGH stand-in's radial displacement: a body with deep folds (concavities: the radius dips to a third), a
    dimple, three thin fins and five spikes -- thin features, concave regions and steep flanks, where a BVH's boxes
    overlap and its depth varies, unlike the smooth blob above."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    r = 1.0 + 0.28 * np.sin(7.0 * x + 0.5) * np.cos(6.0 * z - 0.3) + 0.22 * np.sin(9.0 * y + 1.1) * np.cos(5.0 * x)
    for (cx, cy, cz, amp, sharp) in ((0.0, 0.2, 1.0, -0.55, 14.0),     # dimple facing the camera side
                                     (-0.55, 0.75, 0.15, 0.9, 60.0), (-0.15, 0.96, -0.10, 1.1, 160.0), (0.6, 0.6, 0.5, 1.0, 240.0),
                                     (0.7, -0.1, 0.7, 0.8, 400.0), (-0.8, 0.1, 0.55, 0.9, 300.0)):  # spikes
        c = np.array([cx, cy, cz])
        c = c / np.linalg.norm(c)
        r = r + amp * np.exp(sharp * (d @ c - 1.0))
    for (nx, ny, nz, amp, sharp) in ((1.0, 0.3, 0.0, 0.5, 900.0), (0.0, 1.0, 0.4, 0.45, 700.0), (0.5, 0.0, 1.0, 0.4, 1200.0)):  # fins: great-circle ridges
        nrm = np.array([nx, ny, nz])
        nrm = nrm / np.linalg.norm(nrm)
        r = r + amp * np.exp(-sharp * (d @ nrm) ** 2)
    return np.maximum(r, 0.3)


def make_rough_mesh(n_triangles: int):
    """A second deterministic stand-in with the same triangle count and extents but a hard shape:
    uneven tessellation (a power-law lattice warp: triangle areas spread over a factor of ~50), concavities,
    thin fins and spikes. Returns (vertices float32[V,3], faces int32[n_triangles,3])."""
    n = int(math.ceil(math.sqrt(n_triangles / 12.0)))
    dirs, faces = _cube_sphere(n, warp=lambda t: 0.6 * t + 0.4 * np.sign(t) * np.abs(t) ** 2.4)
    faces = faces[:n_triangles]
    rr = _rough_radius(dirs)
    pos = _CENTER + dirs * _RADII * rr[:, None] / 2.1
    used = np.unique(faces)
    remap = -np.ones(len(pos), np.int64)
    remap[used] = np.arange(len(used))
    return pos[used].astype(np.float32), remap[faces].astype(np.int32)


def write_obj(path, vertices: np.ndarray, faces: np.ndarray, header: str = "") -> None:
    path = Path(path)
    with open(path, "w") as f:
        f.write("# STAND-IN procedural mesh generated by rbrt_amd/standin.py (not the Stanford scan)\n")
        if header:
            f.write(f"# {header}\n")
        np.savetxt(f, vertices, fmt="v %.9g %.9g %.9g")
        np.savetxt(f, faces + 1, fmt="f %d %d %d")


def ensure_obj(path, n_triangles: int, kind: str = "smooth") -> Path:
    """Write the stand-in to `path` unless a file is already there (a real asset wins). kind: smooth | rough."""
    path = Path(path)
    if not path.exists():
        v, f = (make_rough_mesh if kind == "rough" else make_mesh)(n_triangles)
        write_obj(path, v, f, header=f"{n_triangles} triangles, {kind} stand-in")
    return path


def triangles(n_triangles: int, kind: str = "smooth") -> np.ndarray:
    """(N,3,3) float32 triangle soup, as a .obj loader would hand it over."""
    v, f = (make_rough_mesh if kind == "rough" else make_mesh)(n_triangles)
    return v[f]


if __name__ == "__main__":
    import sys
    out = sys.argv[1] if len(sys.argv) > 1 else "bunny.obj"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else BUNNY_TRIANGLES
    ensure_obj(out, n)
    print(f"wrote {out} ({n} triangles, stand-in)")
