"""The GPU-side BVH builder (rbrt_amd/csrc/bvh_device.hip): same contract as the host builder, checked the same way,
plus the end-to-end statement that matters: images and Scene::hit through a GPU-built tree equal the brute-force oracle."""
import ctypes as C
import time

import numpy as np
import pytest

import scenes
from rbrt_amd import abi, standin
from test_bvh_host import check_invariants

pytestmark = pytest.mark.gpu


def build_device(md):
    lib = abi.load_hip()
    nodes, tris = C.c_void_p(), C.c_void_p()
    nn, nt, depth, me, built = C.c_size_t(), C.c_size_t(), C.c_uint32(), C.c_float(), C.c_int()
    abi.check(lib.rbrt_hip_bvh_build_device(C.byref(md.struct), C.byref(nodes), C.byref(nn), C.byref(tris), C.byref(nt),
                                            C.byref(depth), C.byref(me), C.byref(built)))
    if not built.value:
        return None
    N = np.ctypeslib.as_array(C.cast(nodes, C.POINTER(C.c_float)), (nn.value, 32)).copy()
    T = np.ctypeslib.as_array(C.cast(tris, C.POINTER(C.c_float)), (nt.value, 12)).copy()
    lib.rbrt_hip_free_host(nodes)
    lib.rbrt_hip_free_host(tris)
    return N, T, depth.value, me.value


@pytest.fixture(params=["ploc", "lbvh"])
def algo(request, monkeypatch):
    """Both constructions of the binary tree (bvh_device.hip): clustering (default) and the radix tree."""
    monkeypatch.setenv("RBRT_BVH_DEVICE_ALGO", request.param)
    return request.param


@pytest.mark.parametrize("n_tris", [3, 8, 11, 12, 14, 333, 5003, 70003])
def test_device_bvh_invariants(hip, oracle, algo, n_tris):
    md = scenes.standin_mesh(oracle, n_tris, **scenes.EXAMPLE_MESH)
    r = build_device(md)
    n_indexed = sum(1 for i in range((md.n_total // 8) * 8) if not md.is_padding[i])
    if n_indexed <= 4:
        assert r is None  # declined: the host builder's special cases cover these
        return
    check_invariants(md, *r)


def test_device_bvh_adversarial_inputs(hip, oracle, algo):
    """Coincident triangles (every Morton code equal: ties are broken by position), a soup with a few huge triangles,
    non-finite entries (never indexed), a flat mesh (zero extent on one axis)."""
    tri = np.float32([[0, 0, -5], [1, 0, -5], [0, 1, -5]])
    r = build_device(oracle.mesh_prep(np.tile(tri, (3000, 1, 1))))
    assert r is not None
    check_invariants(oracle.mesh_prep(np.tile(tri, (3000, 1, 1))), *r)
    rng = np.random.default_rng(1)
    soup = scenes.random_soup(rng, 4000, extent=3.0, size=0.05)
    soup[::500] *= 40.0
    md = oracle.mesh_prep(soup)
    check_invariants(md, *build_device(md))
    bad = scenes.random_soup(rng, 2000, extent=1.0, size=0.2)
    bad[7, 1, 2] = np.nan
    bad[100, 0, 0] = np.inf
    md = oracle.mesh_prep(bad)
    N, T, depth, me = build_device(md)
    idx = T[:, 9].view(np.uint32)
    assert 7 not in idx and 100 not in idx and len(idx) == 2000 - 2
    flat = scenes.random_soup(rng, 1500, extent=2.0, size=0.3)
    flat[:, :, 1] = 0.25
    md = oracle.mesh_prep(flat)
    check_invariants(md, *build_device(md))


@pytest.mark.parametrize("n_tris", [2003, 20000])
def test_device_built_tree_renders_the_oracle_image(hip, oracle, monkeypatch, algo, n_tris):
    monkeypatch.setenv("RBRT_BVH_BUILDER", "device")
    sc = scenes.example_scene(oracle, n_tris)
    cam = scenes.camera(oracle, 160, 120)
    exp, exp8, _ = oracle.render(cam, sc, abi.default_opts(spp=4, seed=2))
    got, got8 = hip.render_scene(cam, 4, sc, seed=2)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)) and np.array_equal(got8, exp8)
    rng = np.random.default_rng(n_tris)
    md = sc.meshes[0]
    c, R = (md.bbox_lo + md.bbox_hi) / 2, float(np.linalg.norm(md.bbox_hi - md.bbox_lo) / 2)
    o = c + rng.normal(size=(40000, 3)) * R * 2.0
    d = (c + rng.uniform(-1, 1, (40000, 3)) * R) - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.2, 3.0, (40000, 1))
    rays = np.concatenate([o, d], 1).astype(np.float32)
    et, eo, ei, ed = oracle.trace_rays(sc, rays)
    with hip.HipScene(sc) as hs:
        gt, go, gi, gd = hs.trace_rays(rays)
    assert np.array_equal(eo, go) and np.array_equal(ei, gi) and np.array_equal(et.view(np.uint32), gt.view(np.uint32))
    assert (go == 4).sum() > 1000


def test_device_built_config2_frame_equals_the_oracle(hip, oracle, monkeypatch, algo):
    """Config 2's whole frame through a GPU-built tree: the same SHA-256 as through the host-built one and the oracle."""
    import hashlib
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "cfg2_full_1024x768x50_seed1.npz")
    monkeypatch.setenv("RBRT_BVH_BUILDER", "device")
    rad, rgb = hip.render_scene(scenes.camera(oracle, 1024, 768), 50, scenes.example_scene(oracle), seed=1)
    assert hashlib.sha256(rad.tobytes()).hexdigest() == str(g["radiance_sha256"])
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == str(g["rgb8_sha256"])


def test_dragon_sized_scene_setup_time(hip, oracle):
    """scene upload + BVH build of the 871,414-triangle mesh must not dwarf the 5 ms frame (the host
    builder took 0.4 s). Budget: 50 ms; asserted with a margin for a busy box."""
    sc = scenes.example_scene(oracle, standin.DRAGON_TRIANGLES)
    with hip.HipScene(sc):
        pass  # warm-up: first-use costs of the HIP runtime and the code object
    t0 = time.perf_counter()
    with hip.HipScene(sc) as hs:
        dt = time.perf_counter() - t0
        assert hs is not None
    print(f"scene_create, 871,414 triangles: {dt * 1e3:.1f} ms")
    assert dt < 0.15
