"""The tree a scene handle STARTS with and the tree it goes on with (api.cpp struct Refine).

rbrt_hip_scene_create gives a mesh whichever first tree costs the call less -- the device builder's for all but small
meshes -- and a background thread makes the host builder's SAH tree from the records the device builder emitted; the
first render call that finds it on the device adopts it. The reference has no tree at all (mesh.rs:232-243 tests every
triangle), so every tree must give the scan's answer: images and single rays are compared with the oracle before AND
after the switch, and in a stream of frames during which it happens."""
import ctypes as C

import numpy as np
import pytest

import scenes
from rbrt_amd import abi
from test_bvh_host import build


def build_from_records(T):
    lib = abi.load_hip()
    nodes, tris = C.c_void_p(), C.c_void_p()
    nn, nt, depth, me = C.c_size_t(), C.c_size_t(), C.c_uint32(), C.c_float()
    T = np.ascontiguousarray(T, np.float32)
    assert lib.rbrt_hip_bvh_build_host_records(T.ctypes.data_as(C.c_void_p), len(T), C.byref(nodes), C.byref(nn), C.byref(tris),
                                               C.byref(nt), C.byref(depth), C.byref(me)) == 0
    N = np.ctypeslib.as_array(C.cast(nodes, C.POINTER(C.c_float)), (nn.value, 32)).copy()
    R = np.ctypeslib.as_array(C.cast(tris, C.POINTER(C.c_float)), (nt.value, 12)).copy()
    lib.rbrt_hip_free_host(nodes)
    lib.rbrt_hip_free_host(tris)
    return N, R, depth.value, me.value


@pytest.mark.parametrize("n_tris", [5, 12, 333, 5003, 40000])
def test_the_host_builder_makes_the_same_tree_from_records_in_any_order(oracle, n_tris):
    """No GPU. The background thread builds from the device builder's records (leaf order of ANOTHER tree): the result has
    to be, array for array, what the host builder makes of the mesh itself -- every decision of the build is a function of
    sets of triangles, and leaves are ordered by reference index."""
    md = scenes.standin_mesh(oracle, n_tris, **scenes.EXAMPLE_MESH)
    N, T, depth, me = build(md)
    real = T[T[:, 9].view(np.uint32) != 0xFFFFFFFF]
    real = real[np.unique(real[:, 9].view(np.uint32), return_index=True)[1]]  # (one record per triangle: spatial splits duplicate some)
    rng = np.random.default_rng(n_tris)
    for recs in (real, real[::-1], real[rng.permutation(len(real))]):
        N2, T2, depth2, me2 = build_from_records(recs)
        assert depth2 == depth and me2 == me
        assert np.array_equal(N.view(np.uint32), N2.view(np.uint32)) and np.array_equal(T.view(np.uint32), T2.view(np.uint32))


def _rays(sc, rng, n=20000):
    md = sc.meshes[0]
    c, R = (md.bbox_lo + md.bbox_hi) / 2, float(np.linalg.norm(md.bbox_hi - md.bbox_lo) / 2)
    o = c + rng.normal(size=(n, 3)) * R * 2.0
    d = (c + rng.uniform(-1, 1, (n, 3)) * R) - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.2, 3.0, (n, 1))
    return np.concatenate([o, d], 1).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("n_tris", [9003, 20000])
def test_a_handle_renders_the_oracle_image_through_its_first_tree_and_through_the_adopted_one(hip, oracle, monkeypatch, n_tris):
    import torch
    monkeypatch.delenv("RBRT_BVH_BUILDER", raising=False)
    monkeypatch.delenv("RBRT_BVH_REFINE", raising=False)
    sc = scenes.example_scene(oracle, n_tris)
    cam = scenes.camera(oracle, 160, 120)
    opts = abi.default_opts(spp=4, seed=2)
    exp, _, _ = oracle.render(cam, sc, opts)
    rays = _rays(sc, np.random.default_rng(n_tris))
    et, eo, ei, _ = oracle.trace_rays(sc, rays)
    img = torch.empty((120, 160, 3), dtype=torch.float32, device="cuda")
    with hip.HipScene(sc) as hs:
        t = hs.create_times()
        assert t["meshes_device_built"] == 1 and t["meshes_host_built"] == 0  # the cheaper first tree for a mesh of this size
        assert 0.0 < t["hip_init_s"] + t["upload_s"] + t["bvh_build_s"] + t["lanes_s"] <= t["create_s"] * 1.0001
        hs.render_device(cam, opts, img.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(img.cpu().numpy().view(np.uint32), exp.view(np.uint32))
        state, secs = hs.refine_wait(60.0)
        assert state == 1 and secs > 0.0
        info = hs.info()
        assert info["n_meshes_device_built"] == 0  # the host builder's tree is the one in use now
        img.zero_()
        hs.render_device(cam, opts, img.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(img.cpu().numpy().view(np.uint32), exp.view(np.uint32))
        gt, go, gi, _ = hs.trace_rays(rays)
        assert np.array_equal(eo, go) and np.array_equal(ei, gi) and np.array_equal(et.view(np.uint32), gt.view(np.uint32))
        assert hs.refine_wait(0.0)[0] == 1
        hs.check()


@pytest.mark.gpu
def test_the_switch_inside_a_stream_of_frames_changes_no_frame(hip, oracle, monkeypatch):
    """Frames issued back to back while the background build finishes: launches in flight keep the first tree's arrays,
    later ones take the new ones; every frame is the oracle's."""
    import torch
    monkeypatch.delenv("RBRT_BVH_BUILDER", raising=False)
    monkeypatch.delenv("RBRT_BVH_REFINE", raising=False)
    sc = scenes.example_scene(oracle, 60000)
    cam = scenes.camera(oracle, 256, 192)
    opts = abi.default_opts(spp=2, seed=5)
    exp, _, _ = oracle.render(cam, sc, opts, col_stride=8)
    n_frames = 60
    imgs = [torch.empty((192, 256, 3), dtype=torch.float32, device="cuda") for _ in range(n_frames)]
    with hip.HipScene(sc) as hs:
        adopted_at = None
        for k in range(n_frames):
            hs.render_device(cam, opts, imgs[k].data_ptr())
            if adopted_at is None and hs.info()["n_meshes_device_built"] == 0:
                adopted_at = k
        torch.cuda.synchronize()
        state, _ = hs.refine_wait(60.0)
        assert state == 1
        hs.check()
    first = imgs[0].cpu().numpy()
    assert np.array_equal(first[:, ::8].view(np.uint32), exp[:, ::8].view(np.uint32))
    for k in range(1, n_frames):
        assert torch.equal(imgs[k], imgs[0]), f"frame {k} differs (the new trees were adopted at frame {adopted_at})"


@pytest.mark.gpu
def test_refinement_can_be_turned_off_forced_builders_start_none_and_destroy_cancels(hip, oracle, monkeypatch):
    sc = scenes.example_scene(oracle, 20000)
    monkeypatch.delenv("RBRT_BVH_BUILDER", raising=False)  # (the matrix runs force one: this test sets its own)
    monkeypatch.setenv("RBRT_BVH_REFINE", "0")
    with hip.HipScene(sc) as hs:
        assert hs.refine_wait(0.0)[0] == 0 and hs.info()["n_meshes_device_built"] == 1
    monkeypatch.delenv("RBRT_BVH_REFINE", raising=False)
    for forced, n_dev in (("host", 0), ("device", 1)):
        monkeypatch.setenv("RBRT_BVH_BUILDER", forced)
        with hip.HipScene(sc) as hs:
            assert hs.refine_wait(0.0)[0] == 0 and hs.info()["n_meshes_device_built"] == n_dev
    monkeypatch.delenv("RBRT_BVH_BUILDER")
    big = scenes.example_scene(oracle, 200000)
    import time
    for _ in range(5):  # a handle destroyed while its background build is at work: cancelled, nothing waits for the build
        hs = hip.HipScene(big)
        t0 = time.perf_counter()
        hs.close()
        assert time.perf_counter() - t0 < 1.0  # (the 200k-triangle host build alone takes 0.03-0.3 s; a cancelled one ends within a node)
    small = scenes.example_scene(oracle, 83)  # a small mesh: the host builder is the cheaper first tree, nothing follows
    with hip.HipScene(small) as hs:
        assert hs.create_times()["meshes_host_built"] == 1 and hs.refine_wait(0.0)[0] == 0


@pytest.mark.gpu
def test_the_one_shot_call_uses_the_cheaper_builder_and_accounts_for_its_time(hip, oracle, monkeypatch):
    monkeypatch.delenv("RBRT_BVH_BUILDER", raising=False)
    sc = scenes.example_scene(oracle, 20000)
    cam = scenes.camera(oracle, 160, 120)
    exp, exp8, _ = oracle.render(cam, sc, abi.default_opts(spp=4, seed=2))
    got, got8 = hip.render_scene(cam, 4, sc, seed=2)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)) and np.array_equal(got8, exp8)
    t = hip.last_render_times()
    assert t["meshes_device_built"] == 1
    parts = t["create_s"] + t["render_s"] + t["copy_s"] + t["destroy_s"]
    assert 0.0 < parts <= t["total_s"] * 1.0001 and parts >= 0.95 * t["total_s"]


@pytest.mark.gpu
def test_three_meshes_two_of_them_refined_one_small(hip, oracle, monkeypatch):
    """Several meshes in one scene: their records share ONE array (leaf links are absolute positions in it), so the
    background thread relocates every rebuilt tree to its mesh's place; a small mesh between two large ones keeps the host
    builder's first tree and is carried over as it is. Images and single rays before and after the switch."""
    import torch
    monkeypatch.delenv("RBRT_BVH_BUILDER", raising=False)
    monkeypatch.delenv("RBRT_BVH_REFINE", raising=False)
    a = scenes.standin_mesh(oracle, 9000, 45.0, (5.0, -1.8, -12.5), (0, 0, 0), abi.material(abi.MAT_LAMBERTIAN, (0.8, 0.2, 0.2)))
    b = scenes.standin_mesh(oracle, 91, 20.0, (2.5, -1.2, -9.0), (0.1, 0.4, 0.2), abi.material(abi.MAT_DIELECTRIC, (1, 1, 1), 1.5))
    c = scenes.standin_mesh(oracle, 12004, 30.0, (2.0, -1.0, -10.0), (0.3, 0.2, 0.1), abi.material(abi.MAT_METAL, (0.9, 0.9, 0.9), 0.1), kind="rough")
    sc = abi.SceneData(spheres=scenes.EXAMPLE_SPHERES, meshes=[a, b, c])
    cam = scenes.camera(oracle, 160, 120)
    opts = abi.default_opts(spp=4, seed=8)
    exp, _, _ = oracle.render(cam, sc, opts)
    rng = np.random.default_rng(5)
    o = np.float32([3.5, 1.0, -11.0]) + rng.normal(size=(30000, 3)).astype(np.float32) * 6.0
    d = (np.float32([3.5, -1.0, -11.0]) + rng.uniform(-3, 3, (30000, 3)).astype(np.float32)) - o
    rays = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1).astype(np.float32)
    et, eo, ei, _ = oracle.trace_rays(sc, rays)
    img = torch.empty((120, 160, 3), dtype=torch.float32, device="cuda")
    with hip.HipScene(sc) as hs:
        t = hs.create_times()
        assert t["meshes_device_built"] == 2 and t["meshes_host_built"] == 1
        for phase in ("first trees", "after the switch"):
            img.zero_()
            hs.render_device(cam, opts, img.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(img.cpu().numpy().view(np.uint32), exp.view(np.uint32)), phase
            gt, go, gi, _ = hs.trace_rays(rays)
            assert np.array_equal(eo, go) and np.array_equal(ei, gi) and np.array_equal(et.view(np.uint32), gt.view(np.uint32)), phase
            if phase == "first trees":
                assert hs.refine_wait(60.0)[0] == 1 and hs.info()["n_meshes_device_built"] == 0
        hs.check()
    n_el = len(scenes.EXAMPLE_SPHERES)
    assert all((go == n_el + k).sum() > 50 for k in range(3))


@pytest.mark.gpu
def test_handles_made_and_dropped_again_leave_no_device_memory_behind(hip, oracle, monkeypatch):
    """Sixty handles, each with a background build (some adopted, some cancelled by the destroy) and a short stream of frames
    (watcher thread, helper launches in the forced mode on every second handle): free device memory ends where it began."""
    import torch
    monkeypatch.delenv("RBRT_BVH_BUILDER", raising=False)
    sc = scenes.example_scene(oracle, 30000)
    cam = scenes.camera(oracle, 128, 96)
    opts = abi.default_opts(spp=2, seed=1)
    img = torch.empty((96, 128, 3), dtype=torch.float32, device="cuda")
    ref = None

    def once(k):
        nonlocal ref
        monkeypatch.setenv("RBRT_HIP_LAB", "1")
        monkeypatch.setenv("RBRT_HELPERS", "2" if k % 2 else "1")
        with hip.HipScene(sc) as hs:
            if k % 3 == 0:
                hs.refine_wait(30.0)
            for _ in range(4):
                hs.render_device(cam, opts, img.data_ptr())
            torch.cuda.synchronize()
            hs.check()
        got = img.cpu().numpy()
        if ref is None:
            ref = got.copy()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), k

    for k in range(6):  # (the runtime's own pools settle)
        once(k)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(60):
        once(k)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), (free0, free1)
