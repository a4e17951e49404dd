"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerance: BASELINE.json asks for <= 1e-4 per-channel RMSE of the fp32 radiance. The kernel keeps
the reference's operation order and IEEE ops, so these tests assert the much stronger property
that the radiance is BIT-IDENTICAL to the oracle (and fall back to reporting RMSE in the message).
"""
import numpy as np
import pytest

import scenes
from rbrt_amd import abi, standin, tiles

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-4  # north_star: "within 1e-4 per-channel RMSE of the seeded CPU reference"


def assert_same_image(got, exp, what=""):
    assert got.shape == exp.shape
    if np.array_equal(got.view(np.uint32), exp.view(np.uint32)):
        return
    diff = got.astype(np.float64) - exp.astype(np.float64)
    rmse = np.sqrt(np.nanmean(diff ** 2, axis=(0, 1)))
    nbad = int((got.view(np.uint32) != exp.view(np.uint32)).any(axis=-1).sum())
    ys, xs = np.nonzero((got.view(np.uint32) != exp.view(np.uint32)).any(axis=-1))
    first = [(int(y), int(x), got[y, x].tolist(), exp[y, x].tolist()) for y, x in list(zip(ys, xs))[:5]]
    raise AssertionError(f"{what}: {nbad} pixels differ, per-channel RMSE {rmse} (tolerance {RMSE_TOL}); "
                         f"first: {first}")


def test_spheres_only_cfg1_bit_exact(hip, oracle):
    """BASELINE config 1: example_scene.yaml's camera + 4 spheres, 400x300, 8 spp, seed 1."""
    cam = scenes.camera(oracle, 400, 300)
    sc = scenes.spheres_scene()
    opts = abi.default_opts(spp=8, seed=1)
    exp, exp8, _ = oracle.render(cam, sc, opts)
    got, got8 = hip.render_scene(cam, 8, sc, seed=1)
    assert_same_image(got, exp, "cfg1 radiance")
    assert np.array_equal(got8, exp8)


def test_header_spheres_bit_exact(hip, oracle):
    cam = scenes.camera(oracle, 320, 160)
    sc = scenes.spheres_scene(scenes.HEADER_SPHERES)
    opts = abi.default_opts(spp=16, seed=5)
    exp, exp8, _ = oracle.render(cam, sc, opts)
    got, got8 = hip.render_scene(cam, 16, sc, seed=5)
    assert_same_image(got, exp, "header spheres")
    assert np.array_equal(got8, exp8)


@pytest.mark.parametrize("n_tris", [2000, 2003, 2004, 2006])  # N % 8 = 0, 3, 4, 6: all padding cases
@pytest.mark.parametrize("mat", ["dielectric", "lambertian"])
def test_mesh_scene_bit_exact(hip, oracle, n_tris, mat):
    """example_scene.yaml layout with a small stand-in mesh; brute-force oracle vs BVH kernel."""
    over = {} if mat == "dielectric" else {"mat": abi.material(abi.MAT_LAMBERTIAN, (0.9, 0.3, 0.2))}
    sc = scenes.example_scene(oracle, n_tris, mesh_over=over)
    cam = scenes.camera(oracle, 160, 120)
    opts = abi.default_opts(spp=4, seed=2)
    exp, exp8, _ = oracle.render(cam, sc, opts)
    got, got8 = hip.render_scene(cam, 4, sc, seed=2)
    assert_same_image(got, exp, f"mesh {n_tris} {mat}")
    assert np.array_equal(got8, exp8)


def test_mesh_closeup_metal_bit_exact(hip, oracle):
    """Camera pulled close so most rays hit the mesh: stresses traversal, ties and grazing hits."""
    sc = scenes.example_scene(oracle, 5000, mesh_over={"mat": abi.material(abi.MAT_METAL, (0.8, 0.8, 0.8), 0.05)})
    cam = scenes.camera(oracle, 128, 128, position=(5.0, 2.5, -2.0), look_at=(0.0, 0.0, -1.0), up=(0, 1, 0))
    opts = abi.default_opts(spp=4, seed=9)
    exp, _, _ = oracle.render(cam, sc, opts)
    got, _ = hip.render_scene(cam, 4, sc, seed=9)
    assert_same_image(got, exp, "closeup metal")


def _random_rays(rng, n, center, radius):
    o = center + rng.normal(size=(n, 3)) * radius * 2.0
    tgt = center + rng.uniform(-1, 1, (n, 3)) * radius
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= rng.uniform(0.2, 3.0, (n, 1))  # refracted rays are not unit length (dielectric.rs:80-81)
    return np.concatenate([o, d], 1).astype(np.float32)


@pytest.mark.parametrize("n_tris", [1, 7, 8, 9, 777, 20000])
def test_trace_rays_bvh_equals_brute_force(hip, oracle, n_tris):
    """Scene::hit on random rays: (t, object, triangle index, distance) identical to the scan."""
    rng = np.random.default_rng(n_tris)
    sc = scenes.example_scene(oracle, n_tris)
    md = sc.meshes[0]
    center = (md.bbox_lo + md.bbox_hi) / 2
    radius = float(np.linalg.norm(md.bbox_hi - md.bbox_lo) / 2)
    rays = _random_rays(rng, 200_000 if n_tris <= 1000 else 60_000, center, radius)
    et, eo, ei, ed = oracle.trace_rays(sc, rays)
    with hip.HipScene(sc) as hs:
        gt, go, gi, gd = hs.trace_rays(rays)
    assert np.array_equal(eo, go)
    assert np.array_equal(ei, gi)
    assert np.array_equal(et.view(np.uint32), gt.view(np.uint32))
    assert np.array_equal(ed.view(np.uint32), gd.view(np.uint32))
    assert (go >= 4).sum() > 0 or n_tris < 8  # the mesh is actually being hit


def test_trace_rays_degenerate_directions(hip, oracle):
    """Axis-aligned, zero-component, zero and NaN directions: no hang, same answers."""
    sc = scenes.example_scene(oracle, 3000)
    md = sc.meshes[0]
    c = ((md.bbox_lo + md.bbox_hi) / 2).astype(np.float32)
    dirs = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [1, 1, 0], [0, 0, 0],
                     [np.nan, 0, 1], [np.inf, 0, 0], [1e-30, 1, 0], [-0.0, 0.0, -1.0]], np.float32)
    rays = []
    for d in dirs:
        for off in ([0, 0, 0], [0, 0, 20], [0, -20, 0], [20, 0.1, 0.1], [-20, 0, 0]):
            rays.append(np.concatenate([c - np.float32(off), d]))
            rays.append(np.concatenate([c + np.float32(off), d]))
    rays = np.array(rays, np.float32)
    et, eo, ei, ed = oracle.trace_rays(sc, rays)
    with hip.HipScene(sc) as hs:
        gt, go, gi, gd = hs.trace_rays(rays)
    assert np.array_equal(eo, go) and np.array_equal(ei, gi)
    assert np.array_equal(et.view(np.uint32), gt.view(np.uint32))


def test_empty_and_tiny_scenes(hip, oracle):
    cam = scenes.camera(oracle, 64, 48)
    opts = abi.default_opts(spp=2, seed=3)
    empty = abi.SceneData()
    exp, _, _ = oracle.render(cam, empty, opts)
    got, _ = hip.render_scene(cam, 2, empty, seed=3)
    assert_same_image(got, exp, "empty scene")
    one_tri = abi.SceneData(meshes=[oracle.mesh_prep(np.array([[[-3, 0, -8], [3, 0, -8], [0, 4, -8]]], np.float32),
                                                     mat=abi.material(abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5)))])
    exp, _, _ = oracle.render(cam, one_tri, opts)  # N=1: N%8=1 -> the only triangle is truncated away
    got, _ = hip.render_scene(cam, 2, one_tri, seed=3)
    assert_same_image(got, exp, "one (invisible) triangle")
    four = np.tile(np.array([[[-3, 0, -8], [3, 0, -8], [0, 4, -8]]], np.float32), (4, 1, 1))
    four_tri = abi.SceneData(meshes=[oracle.mesh_prep(four, mat=abi.material(abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5)))])
    exp, _, _ = oracle.render(cam, four_tri, opts)  # identical triangles: ties -> lowest index
    got, _ = hip.render_scene(cam, 2, four_tri, seed=3)
    assert_same_image(got, exp, "four identical triangles")
    assert exp.std() > 0


def test_ragged_image_sizes_and_depth_limits(hip, oracle):
    sc = scenes.spheres_scene()
    for (w, h) in ((1, 1), (7, 5), (9, 17), (33, 8)):
        cam = scenes.camera(oracle, w, h)
        for depth in (0, 1, 50, 64):
            opts = abi.default_opts(spp=3, seed=4, max_depth=depth)
            exp, _, _ = oracle.render(cam, sc, opts)
            got, _ = hip.render_scene(cam, 3, sc, seed=4, max_depth=depth)
            assert_same_image(got, exp, f"{w}x{h} depth {depth}")


def test_seed_changes_image_and_is_reproducible(hip, oracle):
    cam = scenes.camera(oracle, 64, 48)
    sc = scenes.spheres_scene()
    a, _ = hip.render_scene(cam, 4, sc, seed=1)
    b, _ = hip.render_scene(cam, 4, sc, seed=1)
    c, _ = hip.render_scene(cam, 4, sc, seed=2)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_tile_sharding_is_partition_invariant(hip, oracle):
    """Rendering rank r of world w and merging gives the single-GPU image bit for bit."""
    cam = scenes.camera(oracle, 100, 60)
    sc = scenes.example_scene(oracle, 1500)
    full, full8 = hip.render_scene(cam, 4, sc, seed=7)
    for world in (2, 3, 8):
        merged = np.full_like(full, np.nan)
        merged8 = np.zeros_like(full8)
        for r in range(world):
            part, part8 = hip.render_scene(cam, 4, sc, seed=7, tile_rank=r, tile_world=world)
            ty, tx = np.meshgrid(np.arange(60) // 8, np.arange(100) // 8, indexing="ij")
            mine = (tiles.tile_number(ty, tx, 13) % world) == r
            merged[mine] = part[mine]
            merged8[mine] = part8[mine]
        assert_same_image(merged, full, f"world {world}")
        assert np.array_equal(merged8, full8)


def test_batched_accumulation_matches_single_batch(hip, oracle, monkeypatch):
    """A tiny workspace forces many sample batches; the per-pixel sum order must not change."""
    cam = scenes.camera(oracle, 96, 64)
    sc = scenes.spheres_scene()
    a, _ = hip.render_scene(cam, 13, sc, seed=11)
    monkeypatch.setenv("RBRT_HIP_WORKSPACE_MB", "1")  # 96*64*12 B = 73 KB/sample -> 14 -> still 1 batch
    cam2 = scenes.camera(oracle, 512, 384)
    b1, _ = hip.render_scene(cam2, 5, sc, seed=11)  # 2.4 MB per sample -> one sample per batch
    monkeypatch.delenv("RBRT_HIP_WORKSPACE_MB")
    b2, _ = hip.render_scene(cam2, 5, sc, seed=11)
    assert np.array_equal(b1, b2)
    exp, _, _ = oracle.render(cam, sc, abi.default_opts(spp=13, seed=11))
    assert_same_image(a, exp, "13 spp")


@pytest.mark.parametrize("depth", [2, 3])
def test_pipelined_launches_match_unpipelined(hip, oracle, monkeypatch, depth):
    """Frame pipeline (rbrt_hip_scene_set_pipeline): sample batches and successive frames overlap on internal
    streams; every frame must still be the oracle's image, whatever is in flight around it."""
    import torch
    cam = scenes.camera(oracle, 160, 120)
    sc = scenes.example_scene(oracle, 2003)
    exp = {seed: oracle.render(cam, sc, abi.default_opts(spp=6, seed=seed))[0] for seed in (3, 4, 5)}
    monkeypatch.setenv("RBRT_HIP_WORKSPACE_MB", "1")  # 160*120*12 B = 230 KB/sample -> 4 samples per batch -> 2 batches
    with hip.HipScene(sc) as hs:
        hs.set_pipeline(depth)
        imgs = [torch.full((120, 160, 3), float("nan"), dtype=torch.float32, device="cuda") for _ in range(6)]
        seeds = [3, 4, 5, 3, 5, 4]
        for img, seed in zip(imgs, seeds):  # six frames queued back to back, no synchronisation in between
            hs.render_device(cam, abi.default_opts(spp=6, seed=seed), img.data_ptr(), None, None)
        torch.cuda.synchronize()
        for k, (img, seed) in enumerate(zip(imgs, seeds)):
            assert_same_image(img.cpu().numpy(), exp[seed], f"pipelined frame {k} (seed {seed})")
        # a counting launch in between runs alone and leaves the pipeline usable
        hs.render_device(cam, abi.default_opts(spp=6, seed=3, flags=abi.FLAG_COLLECT_STATS), imgs[0].data_ptr(), None, None)
        hs.render_device(cam, abi.default_opts(spp=6, seed=4), imgs[1].data_ptr(), None, None)
        torch.cuda.synchronize()
        assert hs.stats()["samples"] == 160 * 120 * 6
        assert_same_image(imgs[0].cpu().numpy(), exp[3], "counting frame")
        assert_same_image(imgs[1].cpu().numpy(), exp[4], "frame after the counting frame")


@pytest.mark.parametrize("env", [{"RBRT_SHARE_IDLE": "0"}, {"RBRT_SHARE_IDLE": "1"}, {"RBRT_SHARE_IDLE": "48"},
                                 {"RBRT_SHARE_IDLE": "1", "RBRT_LDS_STACK": "1"}, {"RBRT_SHARE_BELOW": "0"}])
def test_shared_traversals_do_not_change_the_image(hip, oracle, monkeypatch, env):
    """In the drain idle lanes take over stack entries of busy ones (megakernel.inl, "shared traversal"): whoever
    walks which subtree, the image is the oracle's. Covers the build without sharing (RBRT_SHARE_BELOW=0), sharing
    switched off and at its most eager, and given-away entries that live in the global overflow of the stack."""
    import torch
    cam = scenes.camera(oracle, 96, 64)
    sc = scenes.example_scene(oracle, 3000)
    exp, _, _ = oracle.render(cam, sc, abi.default_opts(spp=3, seed=9))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    out = torch.empty((64, 96, 3), dtype=torch.float32, device="cuda")
    with hip.HipScene(sc) as hs:
        hs.render_device(cam, abi.default_opts(spp=3, seed=9), out.data_ptr())
        torch.cuda.synchronize()
        assert_same_image(out.cpu().numpy(), exp, f"{env}")
        hs.render_device(cam, abi.default_opts(spp=3, seed=9, flags=abi.FLAG_COLLECT_STATS), out.data_ptr())
        torch.cuda.synchronize()
        assert_same_image(out.cpu().numpy(), exp, f"{env} (counting build)")
        given = hs.debug_counters()["shared_entries_given"]
        hs.check()
    sharing = env.get("RBRT_SHARE_IDLE") != "0" and "RBRT_SHARE_BELOW" not in env
    assert (given > 0) == sharing, (env, given)


def test_streamed_and_blocking_frames_interleaved(hip, oracle):
    """The launch policy looks at what is in flight (api.cpp grid_for: half the wave slots for a launch issued while
    another is running, all of them for one that finds the GPU idle, the first launch after a pause still issued as
    one of a stream). Whatever it decides, every frame is the oracle's."""
    import torch
    cam = scenes.camera(oracle, 192, 128)
    sc = scenes.example_scene(oracle, 2003)
    exp = {seed: oracle.render(cam, sc, abi.default_opts(spp=4, seed=seed))[0] for seed in (1, 2)}
    with hip.HipScene(sc) as hs:
        imgs = [torch.full((128, 192, 3), float("nan"), dtype=torch.float32, device="cuda") for _ in range(9)]
        seeds = [1, 2, 1, 2, 2, 1, 1, 2, 1]
        for k in range(4):  # a stream of frames
            hs.render_device(cam, abi.default_opts(spp=4, seed=seeds[k]), imgs[k].data_ptr(), None, None)
        torch.cuda.synchronize()
        for k in (4, 5):  # blocking frames: the first still issued as one of the stream, the second alone
            hs.render_device(cam, abi.default_opts(spp=4, seed=seeds[k]), imgs[k].data_ptr(), None, None)
            torch.cuda.synchronize()
        for k in range(6, 9):  # and a stream again
            hs.render_device(cam, abi.default_opts(spp=4, seed=seeds[k]), imgs[k].data_ptr(), None, None)
        torch.cuda.synchronize()
        hs.check()
    for k, (img, seed) in enumerate(zip(imgs, seeds)):
        assert_same_image(img.cpu().numpy(), exp[seed], f"frame {k} (seed {seed})")


def test_stats_counters(hip, oracle):
    import torch
    cam = scenes.camera(oracle, 64, 48)
    sc = scenes.example_scene(oracle, 3000)
    opts = abi.default_opts(spp=2, seed=1, flags=abi.FLAG_COLLECT_STATS)
    out = torch.empty((48, 64, 3), dtype=torch.float32, device="cuda")
    with hip.HipScene(sc) as hs:
        hs.render_device(cam, opts, out.data_ptr())
        st = hs.stats()
        hs.render_device(cam, abi.default_opts(spp=2, seed=1), out.data_ptr())
        torch.cuda.synchronize()
    exp, _, rays = oracle.render(cam, sc, abi.default_opts(spp=2, seed=1))
    assert st["samples"] == 64 * 48 * 2
    assert st["rays"] == rays
    assert st["nodes_visited"] > 0 and st["tris_tested"] > 0 and st["node_bytes"] == 128
    assert_same_image(out.cpu().numpy(), exp, "render_device")


@pytest.mark.parametrize("w,h,world", [(100, 60, 2), (33, 9, 3), (1024, 768, 8), (64, 64, 1)])
def test_unpack_kernel_matches_numpy_index_math(hip, w, h, world):
    """rbrt_hip_unpack_tiles(_strided) against rbrt_amd/tiles.py (the layout the gloo test exercises)."""
    import torch
    from rbrt_amd import tiles
    rng = np.random.default_rng(5)
    img = rng.random((h, w, 3), dtype=np.float32)
    parts = [tiles.pack(img, r, world) for r in range(world)]
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    out8 = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    tight = torch.from_numpy(np.concatenate(parts).reshape(-1)).cuda()
    hip.unpack_tiles(0, tight.data_ptr(), w, h, world, out.data_ptr(), out8.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), img)
    exp8 = np.clip(np.floor(np.sqrt(img) * np.float32(256.0)), 0, 255).astype(np.uint8)
    assert np.array_equal(out8.cpu().numpy(), exp8)
    stride = hip.packed_pixels(w, h, 0, world) + 64  # any stride >= rank 0's share
    slots = np.zeros((world, stride, 3), np.float32)
    for r in range(world):
        slots[r, :len(parts[r])] = parts[r]
    out.zero_()
    hip.unpack_tiles(0, torch.from_numpy(slots.reshape(-1)).cuda().data_ptr(), w, h, world, out.data_ptr(), None,
                     rank_stride_pixels=stride)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), img)


def test_header_card_scene_bit_exact(hip, oracle):
    """BASELINE config 5's scene (scenes/header_card.yaml: 7 spheres + a lambertian mesh), reduced size."""
    sc = scenes.header_scene(oracle, 3003)
    cam = scenes.camera(oracle, 192, 128)
    exp, exp8, _ = oracle.render(cam, sc, abi.default_opts(spp=6, seed=3))
    got, got8 = hip.render_scene(cam, 6, sc, seed=3)
    assert_same_image(got, exp, "header_card")
    assert np.array_equal(got8, exp8)


def test_two_meshes_and_mesh_order(hip, oracle):
    """Two overlapping meshes: scene.rs:33-41 takes them in YAML order, strictly smaller distance wins."""
    a = scenes.standin_mesh(oracle, 1500, 45.0, (5.0, -1.8, -12.5), (0, 0, 0), abi.material(abi.MAT_LAMBERTIAN, (0.8, 0.2, 0.2)))
    b = scenes.standin_mesh(oracle, 1204, 30.0, (2.0, -1.0, -10.0), (0.3, 0.2, 0.1), abi.material(abi.MAT_METAL, (0.9, 0.9, 0.9), 0.1))
    sc = abi.SceneData(spheres=scenes.EXAMPLE_SPHERES, meshes=[a, b])
    cam = scenes.camera(oracle, 160, 120)
    exp, _, _ = oracle.render(cam, sc, abi.default_opts(spp=4, seed=8))
    got, _ = hip.render_scene(cam, 4, sc, seed=8)
    assert_same_image(got, exp, "two meshes")
    rng = np.random.default_rng(0)
    rays = _random_rays(rng, 50_000, np.float32([3.5, 1.0, -11.0]), 4.0)
    et, eo, ei, ed = oracle.trace_rays(sc, rays)
    with hip.HipScene(sc) as hs:
        gt, go, gi, gd = hs.trace_rays(rays)
    assert np.array_equal(eo, go) and np.array_equal(ei, gi) and np.array_equal(et.view(np.uint32), gt.view(np.uint32))
    assert (go == 4).sum() > 100 and (go == 5).sum() > 100


def test_dragon_sized_mesh_cfg4(hip, oracle):
    """BASELINE config 4: an 871,414-triangle stand-in (deep BVH, stack overflow path). At the example scene's
    scale 45 every triangle of such a fine mesh has |a| < 1e-3 and the reference's test rejects it
    (triangle.rs:198-200: the mesh is invisible, which the kernel reproduces); scale 450 makes it visible."""
    for scale, expect_hits in ((45.0, False), (450.0, True)):
        sc = scenes.example_scene(oracle, standin.DRAGON_TRIANGLES,
                                  mesh_over={"scale": scale, "translation": (5.0 * scale / 45.0, -1.8 * scale / 45.0, -12.5 * scale / 45.0 - 20.0 * (scale > 45))})
        md = sc.meshes[0]
        c = ((md.bbox_lo + md.bbox_hi) / 2).astype(np.float32)
        R = float(np.linalg.norm(md.bbox_hi - md.bbox_lo) / 2)
        rays = _random_rays(np.random.default_rng(4), 6000, c, R)
        et, eo, ei, ed = oracle.trace_rays(sc, rays)
        with hip.HipScene(sc) as hs:
            gt, go, gi, gd = hs.trace_rays(rays)
        assert np.array_equal(eo, go) and np.array_equal(ei, gi)
        assert np.array_equal(et.view(np.uint32), gt.view(np.uint32))
        assert ((go == 4).sum() > 500) == expect_hits
    # full config-4 image size through the megakernel: deterministic and partition-invariant
    cam = scenes.camera(oracle, 1024, 768)
    full, _ = hip.render_scene(cam, 2, sc, seed=1)
    again, _ = hip.render_scene(cam, 2, sc, seed=1)
    assert np.array_equal(full, again)
    part0, _ = hip.render_scene(cam, 2, sc, seed=1, tile_rank=0, tile_world=2)
    part1, _ = hip.render_scene(cam, 2, sc, seed=1, tile_rank=1, tile_world=2)
    ty, tx = np.meshgrid(np.arange(768) // 8, np.arange(1024) // 8, indexing="ij")
    mine = (tiles.tile_number(ty, tx, 128) % 2) == 0
    assert np.array_equal(np.where(mine[..., None], part0, part1), full)
    # and a sub-window of it against the brute-force oracle (108,927 AVX iterations per mesh ray)
    exp, _, _ = oracle.render(cam, sc, abi.default_opts(spp=2, seed=1), window=(500, 532, 300, 316))
    assert np.array_equal(full[300:316, 500:532].view(np.uint32), exp[300:316, 500:532].view(np.uint32))


def test_full_size_cfg2_invariants(hip, oracle, monkeypatch):
    """BASELINE config 2 at full size (1024x768x50, 69,451-triangle stand-in), too big for the brute-force
    oracle: the image must not depend on the pipeline depth, on the sample-batch size, on the tile
    sharding; a strided set of columns is compared with the oracle bit for bit."""
    import hashlib

    import torch
    W, H, SPP = 1024, 768, 50
    cam = scenes.camera(oracle, W, H)
    sc = scenes.example_scene(oracle)

    def sha(t):
        return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()

    img = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    hashes = {}
    with hip.HipScene(sc) as hs:
        for depth in (1, 2, 3, 0):  # 0 = the library's automatic choice
            hs.set_pipeline(depth)
            img.fill_(float("nan"))
            hs.render_device(cam, abi.default_opts(spp=SPP, seed=1), img.data_ptr(), None, None)
            torch.cuda.synchronize()
            hashes[f"pipeline {depth}"] = sha(img)
        ref = img.cpu().numpy().copy()
        monkeypatch.setenv("RBRT_HIP_WORKSPACE_MB", "100")  # 9.4 MB per sample -> 10 samples per launch -> 5 launches
        img.fill_(float("nan"))
        hs.render_device(cam, abi.default_opts(spp=SPP, seed=1), img.data_ptr(), None, None)
        torch.cuda.synchronize()
        hashes["5 pipelined batches"] = sha(img)
        monkeypatch.delenv("RBRT_HIP_WORKSPACE_MB")
        # two ranks' packed tiles, de-interleaved by the unpack kernel
        world = 2
        slot = hip.packed_pixels(W, H, 0, world)
        slots = torch.full((world * slot * 3,), float("nan"), dtype=torch.float32, device="cuda")
        for r in range(world):
            hs.render_device(cam, abi.default_opts(spp=SPP, seed=1, tile_rank=r, tile_world=world),
                             slots[r * slot * 3:].data_ptr(), None, None)
        img.fill_(float("nan"))
        hip.unpack_tiles(0, slots.data_ptr(), W, H, world, img.data_ptr(), None, None, rank_stride_pixels=slot)
        torch.cuda.synchronize()
        hashes["2 ranks merged"] = sha(img)
    assert len(set(hashes.values())) == 1, hashes
    assert not np.isnan(ref).any()
    stride = 64  # 16 columns x 768 rows x 50 spp through the brute-force oracle: a few seconds
    exp, _, _ = oracle.render(cam, sc, abi.default_opts(spp=SPP, seed=1), want_rgb8=False, col_stride=stride)
    assert_same_image(ref[:, ::stride], exp[:, ::stride], "config 2, every 64th column")


@pytest.mark.parametrize("which", ["example_scene.yaml", "header_card.yaml"])
def test_product_host_scene_path_meets_the_kernel(hip, oracle, tmp_path, which):
    """Most GPU tests take the camera and the mesh SoA arrays from the ORACLE's preparation. Here the product's own
    path feeds the kernels: the shipped YAML through the C++ host (yaml_lite, the .obj loader, the transform and the SoA
    conversion of scene.cpp, Camera::create) -> rbrt_scene_t -> HIP render; the expected image comes from the oracle
    on the oracle-prepared scene. Both shipped scenes, ragged image size."""
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    n_tris = 2003
    v, f = standin.make_mesh(n_tris)
    standin.write_obj(tmp_path / "bunny.obj", v, f)
    (tmp_path / "scene.yaml").write_text((root / "scenes" / which).read_text().replace("bunny.obj", str(tmp_path / "bunny.obj")))
    W, H, spp, seed = 150, 92, 5, 12
    hs = abi.HostScene(tmp_path / "scene.yaml", H, W)
    got, got8 = hip.render_scene(hs.camera, spp, hs, seed=seed)
    sc = scenes.example_scene(oracle, n_tris) if which.startswith("example") else scenes.header_scene(oracle, n_tris)
    exp, exp8, _ = oracle.render(scenes.camera(oracle, W, H), sc, abi.default_opts(spp=spp, seed=seed))
    assert_same_image(got, exp, which)
    assert np.array_equal(got8, exp8)


def test_cli_end_to_end_png(hip, oracle, tmp_path):
    """The drop-in path a user of the reference runs: `rbrt -c scene.yaml -t out.png` (YAML -> .obj -> SoA ->
    HIP render -> PNG, C++ host + C ABI). The PNG must hold the oracle's 8-bit image of the same scene and seed."""
    import subprocess
    from pathlib import Path

    from PIL import Image
    root = Path(__file__).resolve().parent.parent
    n_tris = 1203  # N % 8 = 3: the truncated-tail quirk on the way
    v, f = standin.make_mesh(n_tris)
    standin.write_obj(tmp_path / "bunny.obj", v, f)
    text = (root / "scenes" / "example_scene.yaml").read_text().replace("bunny.obj", str(tmp_path / "bunny.obj"))
    (tmp_path / "scene.yaml").write_text(text)
    out = tmp_path / "out.png"
    r = subprocess.run([str(root / "rbrt_amd" / "bin" / "rbrt"), "-c", str(tmp_path / "scene.yaml"), "-t", str(out),
                        "--height", "96", "-w", "128", "-s", "6", "--seed", "9", "--report", str(tmp_path / "rep.json")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Starting rendering" in r.stdout and "100% complete" in r.stdout  # lib.rs:80,112
    # --report: every part of the run named, and on one GPU the parts add up to the total
    import json
    j = json.loads((tmp_path / "rep.json").read_text())
    parts = ("parse_s", "obj_load_s", "prep_s", "hip_init_s", "upload_s", "bvh_build_s", "lanes_s", "buffers_s", "render_s", "gather_s",
             "release_s", "encode_s", "other_s")
    assert all(j[k] >= 0 for k in parts) and abs(sum(j[k] for k in parts) - j["total_s"]) <= 0.02 * j["total_s"] + 1e-4
    assert j["other_s"] <= 0.05 * j["total_s"] + 0.005
    if not __import__("os").environ.get("RBRT_BVH_BUILDER"):
        assert j["bvh_builder"] == "device"  # (1203 entries: the device builder is the cheaper first tree from ~200)
    got = np.array(Image.open(out))
    cam = scenes.camera(oracle, 128, 96)
    sc = scenes.example_scene(oracle, n_tris)
    _, exp8, _ = oracle.render(cam, sc, abi.default_opts(spp=6, seed=9))
    assert got.shape == exp8.shape and np.array_equal(got, exp8)


def test_nan_discriminant_is_reported_like_the_reference_panic(hip, oracle):
    """sphere.rs:33 panics ("Encountered NAN") when a sphere's discriminant is NaN. Across a C ABI that becomes
    a status code: the one-shot call returns RBRT_ERR_NAN, the resident-scene call counts the events."""
    import torch
    bad = list(scenes.EXAMPLE_SPHERES)
    c, r, m = bad[1]
    bad[1] = ((float("nan"), c[1], c[2]), r, m)  # every ray sees a NaN discriminant on this sphere
    sc = scenes.spheres_scene(bad)
    cam = scenes.camera(oracle, 64, 48)
    with pytest.raises(abi.RbrtError) as e:
        hip.render_scene(cam, 2, sc, seed=1)
    assert e.value.code == -6  # RBRT_ERR_NAN
    with hip.HipScene(sc) as hs:
        img = torch.zeros((48, 64, 3), dtype=torch.float32, device="cuda")
        hs.render_device(cam, abi.default_opts(spp=2, seed=1, flags=abi.FLAG_COLLECT_STATS), img.data_ptr(), None, None)
        torch.cuda.synchronize()
        assert hs.stats()["nan_discriminants"] >= 64 * 48 * 2  # at least one per primary ray
        # those rays were treated as misses of that sphere: the image is the scene without it
        exp, _, _ = oracle.render(cam, scenes.spheres_scene([s for i, s in enumerate(scenes.EXAMPLE_SPHERES) if i != 1]),
                                  abi.default_opts(spp=2, seed=1))
        assert_same_image(img.cpu().numpy(), exp, "NaN sphere ignored")


def test_one_scene_many_cameras_and_sizes(hip, oracle):
    """One resident scene, frames of different sizes and spp queued back to back (sample buffers of the pipeline
    lanes grow and are reused, launches of different grids overlap): every frame is the oracle's."""
    import torch
    sc = scenes.example_scene(oracle, 2004)
    jobs = [(64, 48, 3, 1), (200, 120, 5, 2), (64, 48, 3, 3), (33, 17, 9, 4), (200, 120, 2, 5), (8, 8, 1, 6)]
    with hip.HipScene(sc) as hs:
        outs = []
        for w, h, spp, seed in jobs:
            img = torch.full((h, w, 3), float("nan"), dtype=torch.float32, device="cuda")
            hs.render_device(scenes.camera(oracle, w, h), abi.default_opts(spp=spp, seed=seed), img.data_ptr(), None, None)
            outs.append(img)
        torch.cuda.synchronize()
        for (w, h, spp, seed), img in zip(jobs, outs):
            exp, _, _ = oracle.render(scenes.camera(oracle, w, h), sc, abi.default_opts(spp=spp, seed=seed))
            assert_same_image(img.cpu().numpy(), exp, f"{w}x{h}x{spp} seed {seed}")


def test_one_camera_rendered_with_different_distance_windows(hip, oracle):
    """The tile pass grows the tree boxes it tests by a pad that holds 1 / min_dist (the traversal's own), so its tables
    belong to (camera, partition, list order, min_dist): one camera rendered with another window must not be served
    the tables of the first. Blocking frames (which look for a lane that has the camera's tables) and a stream."""
    import torch
    cam = scenes.camera(oracle, 160, 120)
    sc = scenes.example_scene(oracle, 2003)
    windows = [(0.001, 2000.0), (1e-5, 2000.0), (0.05, 50.0), (0.001, 2000.0), (1e-5, 2000.0)]
    with hip.HipScene(sc) as hs:
        outs = []
        for k, (lo, hi) in enumerate(windows):
            img = torch.full((120, 160, 3), float("nan"), dtype=torch.float32, device="cuda")
            hs.render_device(cam, abi.default_opts(spp=3, seed=7, min_dist=lo, max_dist=hi), img.data_ptr(), None, None)
            if k < 3:
                torch.cuda.synchronize()
            outs.append(img)
        torch.cuda.synchronize()
        hs.check()
    for (lo, hi), img in zip(windows, outs):
        exp, _, _ = oracle.render(cam, sc, abi.default_opts(spp=3, seed=7, min_dist=lo, max_dist=hi))
        assert_same_image(img.cpu().numpy(), exp, f"window ({lo}, {hi})")


def test_division_free_mesh_gate_decides_like_the_ieee_form(hip, oracle):
    """The megakernel's mesh gate forms its six slab quotients with v_rcp_f32 and falls back to the verbatim IEEE
    form (aabbox.rs:28-58) only near a decision boundary. Both forms, and the oracle, on random rays, on rays aimed at
    the box's faces, edges and corners to within a few ulp, and on zero / tiny / huge / non-finite components."""
    import ctypes as C
    rng = np.random.default_rng(11)
    lo, hi = np.float32([0.7825403, 0.57846975, -15.222859]), np.float32([7.573573, 7.4303217, -9.879498])  # config 2's mesh box
    n = 400_000
    o = (rng.normal(size=(n, 3)) * 12).astype(np.float32)
    # targets ON the box surface: a random face point, snapped to edges / corners for a third of the rays
    t = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    for k in range(3):
        snap = rng.random(n) < 0.55
        t[snap, k] = np.where(rng.random(snap.sum()) < 0.5, lo[k], hi[k])
    d = (t - o).astype(np.float32)
    d *= rng.choice(np.float32([1e-3, 0.3, 1.0, 1.0, 7.0, 1e4]), (n, 1))
    ulps = rng.integers(-3, 4, (n, 3))
    d = (d.view(np.int32) + ulps.astype(np.int32)).view(np.float32)  # nudge by a few ulp either way
    rays = np.concatenate([o, d], 1)
    # axis-parallel and degenerate directions, origins on / inside / outside the box
    specials = np.float32([0.0, -0.0, 1e-38, -1e-38, 1e-31, 1e31, np.inf, -np.inf, np.nan, 1.0, -1.0])
    extra = []
    for a in specials:
        for b in specials:
            for org in ([0, 0, 0], [4, 4, -12], lo, hi, [lo[0], 4, -12], [4, hi[1], 30], [np.nan, 0, 0], [np.inf, 0, 0]):
                extra.append([*org, a, b, -1.0])
                extra.append([*org, -0.5, a, b])
                extra.append([*org, b, 0.25, a])
    rays = np.concatenate([rays, np.float32(extra)]).astype(np.float32)
    n = len(rays)
    fast, exact = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    abi.check(abi.load_hip().rbrt_hip_selftest_gate(abi.fptr(lo), abi.fptr(hi), abi.fptr(rays), n,
                                                     fast.ctypes.data_as(abi.u8p), exact.ctypes.data_as(abi.u8p)))
    assert np.array_equal(fast, exact), np.flatnonzero(fast != exact)[:10]
    L = oracle.lib()
    sub = np.concatenate([rng.choice(n - len(extra), 20000, replace=False), np.arange(n - len(extra), n)])
    for i in sub:
        assert bool(L.rbrt_oracle_kat_bbox_hit(oracle._p(lo), oracle._p(hi), oracle._p(rays[i]))) == bool(exact[i]), rays[i]
    assert 0.2 < exact.mean() < 0.9


def test_render_pass_checkpoint_and_resume(hip, oracle):
    """rbrt_hip_render_pass: samples in ranges, running sums in a caller-owned buffer. A checkpoint (sums copied to the
    host after 7 of 20 samples) resumed on a NEW scene handle gives the image of one uninterrupted call, bit for bit."""
    import torch
    cam = scenes.camera(oracle, 160, 120)
    sc = scenes.example_scene(oracle, 2003)
    spp, seed = 20, 6
    exp, exp8, _ = oracle.render(cam, sc, abi.default_opts(spp=spp, seed=seed))
    opts = abi.default_opts(spp=spp, seed=seed)
    acc = torch.full((120, 160, 3), float("nan"), dtype=torch.float32, device="cuda")
    with hip.HipScene(sc) as hs:
        hs.render_pass(cam, opts, 0, 3, acc.data_ptr())
        hs.render_pass(cam, opts, 3, 7, acc.data_ptr())
        torch.cuda.synchronize()
        hs.check()
        checkpoint = acc.cpu().numpy().copy()  # what the host writes to disk, with sample_end = 7
    assert not np.isnan(checkpoint).any()
    acc2 = torch.from_numpy(checkpoint).cuda()
    out = torch.full((120, 160, 3), float("nan"), dtype=torch.float32, device="cuda")
    out8 = torch.zeros((120, 160, 3), dtype=torch.uint8, device="cuda")
    with hip.HipScene(sc) as hs:
        hs.render_pass(cam, opts, 7, 8, acc2.data_ptr())
        hs.render_pass(cam, opts, 8, spp, acc2.data_ptr(), out.data_ptr(), out8.data_ptr())
        torch.cuda.synchronize()
        hs.check()
        with pytest.raises(abi.RbrtError):
            hs.render_pass(cam, opts, 5, 5, acc2.data_ptr())
        with pytest.raises(abi.RbrtError):
            hs.render_pass(cam, opts, 5, spp + 1, acc2.data_ptr())
    assert_same_image(out.cpu().numpy(), exp, "resumed render")
    assert np.array_equal(out8.cpu().numpy(), exp8)
    # tile-sharded passes: packed accumulators per rank
    n = hip.packed_pixels(160, 120, 1, 2)
    accp = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    outp = torch.full((n, 3), float("nan"), dtype=torch.float32, device="cuda")
    o2 = abi.default_opts(spp=spp, seed=seed, tile_rank=1, tile_world=2)
    with hip.HipScene(sc) as hs:
        hs.render_pass(cam, o2, 0, 11, accp.data_ptr())
        hs.render_pass(cam, o2, 11, spp, accp.data_ptr(), outp.data_ptr())
        torch.cuda.synchronize()
    from rbrt_amd import tiles
    assert np.array_equal(outp.cpu().numpy().view(np.uint32), tiles.pack(exp, 1, 2).view(np.uint32))


def test_cli_progress_checkpoint_and_resume(hip, oracle, tmp_path):
    """The C++ host end to end with passes: the reference's stdout lines (mesh.rs:29-35 ISA banner, lib.rs:105-110
    progress), a checkpoint left behind by an interrupted run, and a second run that resumes from it: the PNG is
    the oracle's image, the checkpoint is gone."""
    import os
    import subprocess
    from pathlib import Path

    from PIL import Image
    root = Path(__file__).resolve().parent.parent
    v, f = standin.make_mesh(1203)
    standin.write_obj(tmp_path / "bunny.obj", v, f)
    text = (root / "scenes" / "example_scene.yaml").read_text().replace("bunny.obj", str(tmp_path / "bunny.obj"))
    (tmp_path / "scene.yaml").write_text(text)
    out, ck = tmp_path / "out.png", tmp_path / "render.ckpt"
    cmd = [str(root / "rbrt_amd" / "bin" / "rbrt"), "-c", str(tmp_path / "scene.yaml"), "-t", str(out), "--height", "96", "-w", "128",
           "-s", "11", "--seed", "4", "--pass-samples", "3", "--checkpoint", str(ck)]
    r1 = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, RBRT_TEST_STOP_AFTER_PASS="2"))
    assert r1.returncode == 101 and "stopped after pass 2" in r1.stderr
    assert ck.exists() and not out.exists()
    assert "AVX capability detected!" in r1.stdout or "AVX capability not detected" in r1.stdout
    assert "Rendering 27.3% complete!" in r1.stdout and "Rendering 54.5% complete!" in r1.stdout  # 3/11, 6/11
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "Resuming from checkpoint" in r2.stdout and "at sample 6 of 11" in r2.stdout
    assert "Rendering 81.8% complete!" in r2.stdout and "Rendering 100% complete!" in r2.stdout
    assert not ck.exists()
    _, exp8, _ = oracle.render(scenes.camera(oracle, 128, 96), scenes.example_scene(oracle, 1203), abi.default_opts(spp=11, seed=4))
    assert np.array_equal(np.array(Image.open(out)), exp8)
    # a checkpoint of a different render is not resumed
    r3 = subprocess.run(cmd[:-2] + ["--checkpoint", str(ck), "--seed", "5"], capture_output=True, text=True, timeout=300,
                        env=dict(os.environ, RBRT_TEST_STOP_AFTER_PASS="1"))
    assert r3.returncode == 101 and ck.exists()
    r4 = subprocess.run(cmd, capture_output=True, text=True, timeout=300)  # seed 4 again: must start over
    assert r4.returncode == 0 and "does not match this render" in r4.stdout
    assert np.array_equal(np.array(Image.open(out)), exp8)


def test_short_ieee_forms_match_the_compilers(hip):
    """The kernels take hipcc's correctly rounded sqrt and division without their range wrapping when a wave's
    operands are all in the everyday range, sharing the reciprocal refinement between the three divisions of a
    normalize. 2^28 pseudo-random operands (three quarters in range, one quarter with zeros, denormals, huge values,
    inf and NaN mixed in): not one bit of difference from the compiler's forms."""
    import ctypes as C
    counts = (C.c_uint64 * 3)()
    total_fast = 0
    for seed in (1, 2):
        abi.check(abi.load_hip().rbrt_hip_selftest_ieee(seed, 1 << 27, counts))
        assert counts[0] == 0 and counts[1] == 0, list(counts)
        total_fast += counts[2]
    assert total_fast == 2 * 3 * (1 << 27) // 4  # the in-range waves really ran the short forms' domain


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("RBRT_FUZZ_SCENES", "16"))))
def test_random_scenes_bit_exact(hip, oracle, seed):
    """Randomised scenes: 0-9 spheres, on every third seed 1-6 BasicTriangle elements shuffled in between them, and
    0-3 overlapping triangle soups / stand-in meshes with random materials and
    transforms, ragged image sizes, random depth limits. Exercises mesh order, ties between meshes and spheres, the
    per-scene triangle array with absolute leaf links, and both BVH builders (small meshes forced to the GPU builder
    on odd seeds)."""
    import os
    rng = np.random.default_rng(1000 + seed)
    kinds = [abi.MAT_LAMBERTIAN, abi.MAT_METAL, abi.MAT_DIELECTRIC]

    def rand_mat():
        k = kinds[int(rng.integers(3))]
        return abi.material(k, tuple(rng.uniform(0.05, 0.95, 3)), float(rng.uniform(0.0, 0.6) if k == abi.MAT_METAL else rng.uniform(0.3, 2.2)))
    spheres = [((0.0, -1000.0, -5.0), 1000.0, rand_mat())] if rng.random() < 0.7 else []
    for _ in range(int(rng.integers(0, 9))):
        spheres.append((tuple(rng.uniform(-6, 6, 3) + np.array([0, 2, -10])), float(rng.uniform(0.3, 2.5)), rand_mat()))
    meshes = []
    for _ in range(int(rng.integers(0, 4))):
        n = int(rng.integers(1, 700))
        if rng.random() < 0.5:
            tri = scenes.random_soup(rng, n, extent=float(rng.uniform(0.5, 3.0)), size=float(rng.uniform(0.05, 1.0)))
            meshes.append(oracle.mesh_prep(tri, float(rng.uniform(0.5, 2.0)), tuple(rng.uniform(-1, 1, 3)),
                                           tuple(rng.uniform(-3, 3, 3) + np.array([0, 1.5, -9])), rand_mat()))
        else:
            meshes.append(scenes.standin_mesh(oracle, n + 50, float(rng.uniform(15, 60)), tuple(rng.uniform(-4, 4, 3) + np.array([0, 0, -10])),
                                              tuple(rng.uniform(-1, 1, 3)), rand_mat()))
    # (drawn from a second generator, so that the scenes of the seeds that have no triangles stay what they were)
    rng_t = np.random.default_rng(5000 + seed)
    tris, order = [], None
    if seed % 3 == 0:  # BasicTriangle elements (triangle.rs:9-34), shuffled in between the spheres
        for _ in range(int(rng_t.integers(1, 7))):
            c = rng_t.uniform(-5, 5, 3) + np.array([0, 2, -9])
            tris.append((tuple(map(tuple, c + rng_t.uniform(-3, 3, (3, 3)))), rand_mat()))
        order = [i for i in range(len(spheres))] + [scenes.T | i for i in range(len(tris))]
        rng_t.shuffle(order)
    sc = abi.SceneData(spheres=spheres, meshes=meshes, triangles=tris, element_order=order)
    w, h = int(rng.integers(20, 90)), int(rng.integers(20, 70))
    # (round 4: not the position alone -- the camera looks at a random point of the scene's middle, with an `up` that is
    # not orthogonal to that direction (cam.rs:30-33, 55 takes whatever it is given) and a random focal length; drawn from a
    # generator of its own, so that the scenes of the seeds stay what they were)
    rng_c = np.random.default_rng(9000 + seed)
    pos = rng.uniform(-2, 2, 3) + np.array([0, 4, 4])
    look = (rng_c.uniform(-4, 4, 3) + np.array([0, 1.5, -10])) - pos
    up = np.array([0.0, 1.0, 0.0]) + rng_c.uniform(-0.5, 0.5, 3)
    cam = scenes.camera(oracle, w, h, position=tuple(pos), look_at=tuple(look / np.linalg.norm(look)), up=tuple(up),
                        focal_mm=float(np.exp(rng_c.uniform(np.log(12.0), np.log(120.0)))))
    spp, depth = int(rng.integers(1, 5)), int(rng.choice([1, 3, 50]))
    old = os.environ.get("RBRT_BVH_BUILDER")
    if seed % 2:
        os.environ["RBRT_BVH_BUILDER"] = "device"
    try:
        exp, exp8, _ = oracle.render(cam, sc, abi.default_opts(spp=spp, seed=seed, max_depth=depth))
        got, got8 = hip.render_scene(cam, spp, sc, seed=seed, max_depth=depth)
    finally:
        if old is None:
            os.environ.pop("RBRT_BVH_BUILDER", None)
        else:
            os.environ["RBRT_BVH_BUILDER"] = old
    assert_same_image(got, exp, f"random scene {seed}: {len(spheres)} spheres, {[m.n_real for m in meshes]} triangles, {w}x{h}x{spp}, depth {depth}")
    assert np.array_equal(got8, exp8)
