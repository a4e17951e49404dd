"""Host-side BVH builder (no GPU): structural invariants of the 4-wide tree, and that culling through it
can never lose the triangle the reference's brute-force scan returns."""
import ctypes as C

import numpy as np
import pytest

import scenes
from rbrt_amd import abi

NO_CHILD = -2 ** 31
LEAF_BITS = 2  # device_types.h kLeafBits (RBRT_LEAF_BITS)
LEAF_MAX = 1 << LEAF_BITS


def build(md):
    lib = abi.load_hip()
    nodes, tris = C.c_void_p(), C.c_void_p()
    nn, nt, depth, me = C.c_size_t(), C.c_size_t(), C.c_uint32(), C.c_float()
    assert lib.rbrt_hip_bvh_build_host(C.byref(md.struct), C.byref(nodes), C.byref(nn), C.byref(tris), C.byref(nt),
                                       C.byref(depth), C.byref(me)) == 0
    N = np.ctypeslib.as_array(C.cast(nodes, C.POINTER(C.c_float)), (nn.value, 32)).copy()
    T = np.ctypeslib.as_array(C.cast(tris, C.POINTER(C.c_float)), (nt.value, 12)).copy()
    lib.rbrt_hip_free_host(nodes)
    lib.rbrt_hip_free_host(tris)
    return N, T, depth.value, me.value


def tri_boxes(T):
    v0, e1, e2 = T[:, 0:3], T[:, 3:6], T[:, 6:9]
    pts = np.stack([v0, v0 + e1, v0 + e2], 1)
    return pts.min(1), pts.max(1)


@pytest.mark.parametrize("n_tris", [0, 1, 5, 8, 11, 12, 14, 333, 5003])
def test_bvh4_invariants(oracle, n_tris):
    if n_tris == 0:
        md = oracle.mesh_prep(np.zeros((0, 3, 3), np.float32))
    else:
        md = scenes.standin_mesh(oracle, n_tris, **scenes.EXAMPLE_MESH)
    check_invariants(md, *build(md))


def test_bvh4_invariants_with_spatial_splits(oracle):
    """Meshes where the builder does cut triangles: the rough stand-in (fins, spikes, folds) and a cloud of small triangles
    with long needles through it. Boxes nest, parts cover their triangles, duplicates stay inside the budget."""
    md = scenes.standin_mesh(oracle, 6003, kind="rough", **scenes.EXAMPLE_MESH)
    N, T, depth, me = build(md)
    assert len(T) > 6003  # some references are duplicated
    check_invariants(md, N, T, depth, me)
    rng = np.random.default_rng(11)
    a = rng.uniform(-1, 1, (120, 3))
    dirs = rng.normal(size=(120, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    needles = np.stack([a, a + dirs * rng.uniform(0.5, 2.0, (120, 1)), a + dirs * 0.01 + rng.normal(size=(120, 3)) * 0.02], 1)
    soup = np.concatenate([scenes.random_soup(rng, 1880, extent=1.0, size=0.04), needles.astype(np.float32)])
    md = oracle.mesh_prep(soup[rng.permutation(len(soup))])
    N, T, depth, me = build(md)
    assert len(T) > 2000 + 100  # the needles are cut many times
    check_invariants(md, N, T, depth, me)


def check_invariants(md, N, T, depth, max_e12):
    """What every builder (host bvh.cpp, GPU bvh_device.hip) owes the traversal: bvh.h's contract. The host builder may
    reference a triangle from several leaves (spatial splits), each with the box of the PART of the triangle it covers:
    boxes nest, and the parts' boxes together cover the triangle."""
    import sys
    sys.setrecursionlimit(10000)
    child = N[:, 24:28].view(np.int32)
    n_tested = (md.n_total // 8) * 8
    want = [i for i in range(n_tested) if not md.is_padding[i]]   # triangle.rs:166-167, :400
    idx = T[:, 9].view(np.uint32)
    real = idx != 0xFFFFFFFF
    assert sorted(set(idx[real].tolist())) == want                 # every returnable triangle, and nothing else
    assert real.sum() <= len(want) + int(0.60 * len(want)) + 1     # duplicated references within the budget (bvh.h kSpatialBudget)
    for i in np.nonzero(real)[0]:                                  # records are bit copies of the SoA streams
        j = idx[i]
        assert T[i, 0] == md.arrays["v0x"][j] and T[i, 4] == md.arrays["e1y"][j] and T[i, 8] == md.arrays["e2z"][j]
    e12 = np.linalg.norm(T[:, 3:6], axis=1) * np.linalg.norm(T[:, 6:9], axis=1)
    assert depth <= 20 and len(N) >= 1
    leaf_boxes = {}  # reference index -> boxes of the leaves that hold it

    def check(node, d):
        """returns (lo, hi, e12) of everything below `node`: the union of its stored child boxes, which the parent's box
        of this node has to contain"""
        assert d <= depth
        lo_all, hi_all, e_all = np.full(3, np.inf), np.full(3, -np.inf), 0.0
        for k in range(4):
            c = child[node, k]
            if c == NO_CHILD:
                assert (N[node, [k, 4 + k, 8 + k]] == np.inf).all() and (N[node, [12 + k, 16 + k, 20 + k]] == -np.inf).all()  # the empty box
                continue
            blo = N[node, [k, 4 + k, 8 + k]]
            bhi = N[node, [12 + k, 16 + k, 20 + k]]
            if c >= 0:
                lo, hi, e = check(c, d + 1)
                assert (blo <= lo).all() and (bhi >= hi).all()
            else:
                first, cnt = (~c) >> LEAF_BITS, ((~c) & (LEAF_MAX - 1)) + 1
                e = e12[first:first + cnt].max()
                assert (np.diff(idx[first:first + cnt].astype(np.int64)) > 0).all()  # ascending index inside a leaf
                for j in idx[first:first + cnt]:
                    if j != 0xFFFFFFFF:
                        leaf_boxes.setdefault(int(j), []).append((blo.astype(np.float64), bhi.astype(np.float64)))
            assert N[node, 28 + k] >= e * (1 - 1e-6)
            lo_all, hi_all, e_all = np.minimum(lo_all, blo), np.maximum(hi_all, bhi), max(e_all, N[node, 28 + k])
        return lo_all, hi_all, e_all

    check(0, 0)
    # coverage: points of every triangle (corners, edge points, interior points) lie in the box of a leaf that references it
    first_rec = {}
    for i in np.nonzero(real)[0]:
        first_rec.setdefault(int(idx[i]), i)
    bary = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [.5, .5, 0], [0, .5, .5], [.5, 0, .5], [1 / 3, 1 / 3, 1 / 3], [.8, .1, .1], [.1, .8, .1],
                     [.1, .1, .8], [.25, .75, 0], [0, .25, .75], [.75, 0, .25], [.6, .3, .1], [.05, .5, .45]])
    for j, boxes in leaf_boxes.items():
        r = T[first_rec[j]]
        v0 = r[0:3].astype(np.float32)
        v1, v2 = (v0 + r[3:6].astype(np.float32)).astype(np.float64), (v0 + r[6:9].astype(np.float32)).astype(np.float64)
        pts = bary[:, :1] * v0.astype(np.float64) + bary[:, 1:2] * v1 + bary[:, 2:3] * v2
        tol = 1e-5 * (np.abs(pts).max() + 1e-30)
        inside = np.zeros(len(pts), bool)
        for (lo, hi) in boxes:
            inside |= ((pts >= lo - tol) & (pts <= hi + tol)).all(1)
        assert inside.all(), (j, len(boxes))


@pytest.mark.parametrize("kind", ["smooth", "rough"])
def test_bvh_culling_keeps_the_brute_force_winner(oracle, kind):
    """CPU walk of the BVH with the kernel's pad formula: the scan's winning triangle (oracle) is always
    among the triangles of the leaves the walk reaches, for rays from near and far, unit and non-unit."""
    sc = scenes.example_scene(oracle, 3001, kind=kind)
    md = sc.meshes[0]
    N, T, _, _ = build(md)
    if kind == "rough":  # (the case spatial splits are for: some references must have been duplicated)
        assert len(T) > len(np.unique(T[:, 9].view(np.uint32)))
    child = N[:, 24:28].view(np.int32)
    idx = T[:, 9].view(np.uint32)
    rng = np.random.default_rng(3)
    c = ((md.bbox_lo + md.bbox_hi) / 2).astype(np.float64)
    R = float(np.linalg.norm(md.bbox_hi - md.bbox_lo) / 2)
    n = 1500
    o = c + rng.normal(size=(n, 3)) * R * rng.choice([1.5, 4.0, 60.0], (n, 1))
    d = (c + rng.uniform(-1, 1, (n, 3)) * R * 0.9) - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.2, 3.0, (n, 1))
    rays = np.concatenate([o, d], 1).astype(np.float32)
    only_mesh = abi.SceneData(meshes=[md])
    t, obj, tri, _ = oracle.trace_rays(only_mesh, rays)
    assert (obj >= 0).sum() > 300
    eps, radius = 0.001, np.linalg.norm((md.bbox_hi - md.bbox_lo).astype(np.float64) / 2) * 1.0001
    for r in np.nonzero(obj >= 0)[0]:
        oo, dd = rays[r, :3].astype(np.float64), rays[r, 3:].astype(np.float64)
        pad_base = 64 * 2.0 ** -24 * (np.linalg.norm(oo - c) + radius + np.abs(oo).max())
        pad_k = pad_base * np.linalg.norm(dd) / eps
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / dd
        stack, reached = [0], set()
        while stack:
            node = stack.pop()
            if node < 0:
                first, cnt = (~node) >> LEAF_BITS, ((~node) & (LEAF_MAX - 1)) + 1
                reached.update(idx[first:first + cnt].tolist())
                continue
            for k in range(4):
                if child[node, k] == NO_CHILD:
                    continue
                pad = pad_base + pad_k * N[node, 28 + k]
                lo = N[node, [k, 4 + k, 8 + k]].astype(np.float64) - pad
                hi = N[node, [12 + k, 16 + k, 20 + k]].astype(np.float64) + pad
                with np.errstate(invalid="ignore"):
                    t0, t1 = (lo - oo) * inv, (hi - oo) * inv
                tn, tf = np.fmax.reduce(np.fmin(t0, t1)), np.fmin.reduce(np.fmax(t0, t1))
                if tn <= tf and tf >= eps and tn <= t[r]:  # pruned with the FINAL best t: the hardest case
                    stack.append(int(child[node, k]))
        assert int(tri[r]) in reached, (r, tri[r])


def test_threaded_build_equals_single_threaded(oracle, monkeypatch):
    """Subtrees built by worker threads are spliced back in DFS order: same arrays for any thread count."""
    md = scenes.standin_mesh(oracle, 70003, **scenes.EXAMPLE_MESH)  # above the 32768-triangle threading threshold
    monkeypatch.setenv("RBRT_BVH_THREADS", "1")
    N1, T1, d1, e1 = build(md)
    for threads in ("2", "5"):
        monkeypatch.setenv("RBRT_BVH_THREADS", threads)
        N, T, d, e = build(md)
        assert d == d1 and e == e1
        assert np.array_equal(N.view(np.uint32), N1.view(np.uint32)) and np.array_equal(T.view(np.uint32), T1.view(np.uint32))


def test_depth_budget_adversarial_and_oversize(oracle):
    """The 2-bit leaf count field holds at most 4 triangles. (1) 3000 coincident triangles (no spatial split exists:
    the builder must fall back to median splits and still end with <= 4 per leaf inside the depth budget);
    (2) a mesh above the 8,388,608 triangles the depth budget always accommodates is refused up front."""
    tri = np.float32([[0, 0, -5], [1, 0, -5], [0, 1, -5]])
    md = oracle.mesh_prep(np.tile(tri, (3000, 1, 1)))
    N, T, depth, _ = build(md)
    child = N[:, 24:28].view(np.int32)
    leaves = child[(child < 0) & (child != NO_CHILD)]
    counts = ((~leaves) & (LEAF_MAX - 1)) + 1
    firsts = (~leaves) >> LEAF_BITS
    assert counts.max() <= LEAF_MAX and depth <= 20
    assert (firsts + counts).max() <= len(T)
    assert counts.sum() == len(T)  # every record in exactly one leaf
    big = np.zeros(8, np.float32)  # the size check comes before any array is read
    m = abi.Mesh()
    m.n_total = (4 << 21) + 8
    m.n_real = m.n_total
    for k in abi.MeshData.FIELDS:
        setattr(m, k, abi.fptr(big))
    m.is_padding = np.zeros(8, np.uint8).ctypes.data_as(abi.u8p)
    m.mat = abi.material(abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))
    sc = abi.Scene(0, None, 1, C.pointer(m))
    h = C.c_void_p()
    assert abi.load_hip().rbrt_hip_scene_create(C.byref(sc), 0, C.byref(h)) == abi.RBRT_ERR_UNSUPPORTED
    assert b"8,388,608" in abi.load_hip().rbrt_hip_last_error()
