#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (oracle/rbrt_oracle.cpp).

    python tests/golden/make_golden.py

The reference itself cannot run here (Rust, no toolchain; and it has no RNG seed), so the goldens
are outputs of the build's own seeded restatement, which tests/test_oracle_kats.py pins against the
reference's unit-test vectors. Each fixture = inputs (scene description by name + seed + size) and
expected outputs: SHA-256 of the full fp32 radiance / RGB8 bytes plus a sub-sampled copy of the
radiance for diagnosing a mismatch. No reference source text is stored here.
"""
import hashlib
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))
from oracle import pyoracle  # noqa: E402
import scenes  # noqa: E402
from rbrt_amd import abi  # noqa: E402

CASES = {
    # BASELINE config 1: camera + 4 spheres of example_scene.yaml, no mesh, 400x300, 8 spp, seed 1
    "cfg1_spheres_400x300x8_seed1": dict(kind="spheres", w=400, h=300, spp=8, seed=1),
    "header_spheres_320x160x16_seed5": dict(kind="header_spheres", w=320, h=160, spp=16, seed=5),
    # example_scene.yaml layout with small stand-in meshes covering the padding cases N % 8 = 0,3,4,6
    "mesh2000_160x120x4_seed2": dict(kind="mesh", n=2000, w=160, h=120, spp=4, seed=2),
    "mesh2003_160x120x4_seed2": dict(kind="mesh", n=2003, w=160, h=120, spp=4, seed=2),
    "mesh2004_160x120x4_seed2": dict(kind="mesh", n=2004, w=160, h=120, spp=4, seed=2),
    "mesh2006_160x120x4_seed2": dict(kind="mesh", n=2006, w=160, h=120, spp=4, seed=2),
    # BasicTriangle elements (triangle.rs:9-34) interleaved with the example spheres, without and with a mesh behind them
    "triangles_200x150x8_seed3": dict(kind="triangles", n=0, w=200, h=150, spp=8, seed=3),
    "triangles_mesh1203_160x120x4_seed4": dict(kind="triangles", n=1203, w=160, h=120, spp=4, seed=4),
}


def build_scene(case):
    if case["kind"] == "spheres":
        return scenes.spheres_scene()
    if case["kind"] == "header_spheres":
        return scenes.spheres_scene(scenes.HEADER_SPHERES)
    if case["kind"] == "triangles":
        return scenes.triangle_scene(pyoracle, case["n"])
    return scenes.example_scene(pyoracle, case["n"])


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main(only=None):
    for name, case in CASES.items():
        if only and only not in name:
            continue
        cam = scenes.camera(pyoracle, case["w"], case["h"])
        sc = build_scene(case)
        rad, rgb, rays = pyoracle.render(cam, sc, abi.default_opts(spp=case["spp"], seed=case["seed"]))
        np.savez_compressed(HERE / f"{name}.npz", radiance_sha256=sha(rad), rgb8_sha256=sha(rgb),
                            radiance_sub=rad[::4, ::4].copy(), rgb8_sub=rgb[::4, ::4].copy(), rays=np.int64(rays),
                            mean=rad.mean(axis=(0, 1)))
        print(name, sha(rad)[:16], rays)
    if only:
        return
    # single-ray records: Scene::hit on a fixed bundle of rays through the example scene + 3001-triangle stand-in
    sc = scenes.example_scene(pyoracle, 3001)
    md = sc.meshes[0]
    rng = np.random.default_rng(12345)
    c = (md.bbox_lo + md.bbox_hi) / 2
    R = float(np.linalg.norm(md.bbox_hi - md.bbox_lo) / 2)
    o = np.float32([0.0, 5.0, 4.0]) + rng.normal(size=(600, 3)) * 0.5
    tgt = np.where(rng.random((600, 1)) < 0.6, c + rng.uniform(-1, 1, (600, 3)) * R,
                   np.float32([0, 1, -10]) + rng.uniform(-8, 8, (600, 3)))
    d = tgt - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.3, 2.0, (600, 1))
    rays = np.concatenate([o, d], 1).astype(np.float32)
    t, obj, tri, dist = pyoracle.trace_rays(sc, rays)
    np.savez_compressed(HERE / "rays_example3001.npz", rays=rays, t=t, obj=obj, tri=tri, dist=dist)
    print("rays", int((obj >= 4).sum()), "mesh hits,", int((obj >= 0).sum()), "hits of 600")


# (material kind, parameter): lambertian, a polished and a rough metal, the glass sphere's index and the example mesh's
SCATTER_MATERIALS = [(0, 0.0), (1, 0.005), (1, 0.6), (2, 1.8), (2, 0.2)]


def make_scatter_events():
    """SURVEY 8(c) fixture 4, second half: single scatter events (materials.rs:4-12) -> attenuation, next ray, bool,
    with explicit random streams: 300 per material. A future image mismatch can be localised to M3 / M4 / M5 by
    replaying these through rbrt_hip_debug_scatter instead of bisecting images."""
    import ctypes as C
    rng = np.random.default_rng(2024)
    n_per = 300
    n = n_per * len(SCATTER_MATERIALS)
    kind, param = np.zeros(n, np.int32), np.zeros(n, np.float32)
    albedo = np.zeros((n, 3), np.float32)
    in_ray, point, normal = np.zeros((n, 6), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    att, out_ray, ok = np.zeros((n, 3), np.float32), np.zeros((n, 6), np.float32), np.zeros(n, np.uint8)
    before, after = np.zeros((n, 2), np.uint32), np.zeros((n, 2), np.uint32)
    key = np.zeros((n, 3), np.uint64)
    u32p = C.POINTER(C.c_uint32)
    for i in range(n):
        k, prm = SCATTER_MATERIALS[i // n_per]
        kind[i], param[i] = k, prm
        albedo[i] = rng.uniform(0.05, 0.95, 3)
        d = rng.uniform(-1, 1, 3) * rng.uniform(0.3, 2)
        in_ray[i] = np.concatenate([rng.uniform(-10, 10, 3), d])
        point[i] = rng.uniform(-10, 10, 3)
        # sphere normals come unnormalised (sphere.rs:56: up to the ground sphere's radius 1000), mesh normals unit
        nrm = rng.uniform(-1, 1, 3)
        normal[i] = nrm / np.linalg.norm(nrm) if i % 3 == 0 else nrm * rng.uniform(0.3, 1000)
        if i % 7 == 0:  # a normal on the ray's side / against it, exactly aligned: the sign tests' edge
            normal[i] = (d if i % 14 == 0 else -d).astype(np.float32)
        key[i] = (int(rng.integers(1 << 40)), int(rng.integers(1 << 20)), int(rng.integers(4096)))
        mat = abi.material(int(k), tuple(float(x) for x in albedo[i]), float(prm))
        ok[i] = pyoracle.lib().rbrt_oracle_kat_scatter_state(
            C.byref(mat), pyoracle._p(in_ray[i]), pyoracle._p(point[i]), pyoracle._p(normal[i]), int(key[i, 0]), int(key[i, 1]),
            int(key[i, 2]), pyoracle._p(att[i]), pyoracle._p(out_ray[i]), before[i].ctypes.data_as(u32p), after[i].ctypes.data_as(u32p))
    np.savez_compressed(HERE / "scatter_events.npz", kind=kind, param=param, albedo=albedo, in_ray=in_ray, point=point, normal=normal,
                        stream_key=key, state_before=before, attenuation=att, out_ray=out_ray, ok=ok, state_after=after)
    print("scatter events", n, "false:", int((ok == 0).sum()))


if __name__ == "__main__":
    if sys.argv[1:] == ["scatter"]:
        make_scatter_events()
    elif sys.argv[1:2] == ["only"]:
        main(sys.argv[2])
    else:
        main()
        make_scatter_events()
