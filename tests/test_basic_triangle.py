"""BasicTriangle (rbrt_lib/src/triangle.rs:9-34, 92-130, 412-441): the reference's second `Intersectable`, a single
triangle as an element of Scene::elements. The YAML factory never builds one (blueprints.rs:132-158), the scene type
admits it, so the C ABI carries it (rbrt_scene_t::triangles / element_order) and the kernels test it in element order.

not gpu: the oracle's restatement against the independent numpy one -- single rays incl. the boundary cases of every
         test in the routine, and whole pixels of a scene that mixes spheres, triangles and a mesh.
gpu    : Scene::hit and whole images through the C ABI against the oracle, bit for bit; ties between coincident
         elements go to the earlier one in Scene::elements order; a malformed order is refused.
(The committed images are in tests/test_golden.py: triangles_200x150x8_seed3, triangles_mesh1203_160x120x4_seed4.)
"""
import ctypes as C

import numpy as np
import pytest

import np_reference as R
import scenes
from rbrt_amd import abi

f32 = np.float32


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _oracle_tri(oracle, corners, mat, ray, min_dist=0.001, max_dist=2000.0):
    tri = abi.Triangle()
    for a in range(3):
        for b in range(3):
            tri.corners[a][b] = float(corners[a][b])
    tri.mat = mat
    t, d, n = np.zeros(1, f32), np.zeros(1, f32), np.zeros(3, f32)
    ok = oracle.lib().rbrt_oracle_kat_basic_triangle(C.byref(tri), oracle._p(np.ascontiguousarray(ray, f32)), min_dist, max_dist,
                                                    oracle._p(t), oracle._p(d), oracle._p(n))
    return (bool(ok), t[0], d[0], n)


def test_single_rays_oracle_equals_numpy(oracle):
    rng = np.random.default_rng(77)
    mat = abi.material(abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))
    n_hit = 0
    for k in range(1500):
        corners = rng.uniform(-3, 3, (3, 3)).astype(f32)
        if k % 10 == 0:   # tiny triangle: |a| < eps rejects it from most directions (triangle.rs:99-101)
            corners = (corners[0] + rng.uniform(-0.02, 0.02, (3, 3))).astype(f32)
        o = rng.uniform(-6, 6, 3).astype(f32)
        w = rng.dirichlet((1, 1, 1)) if k % 3 else rng.dirichlet((1, 1, 1)) * np.array([2.0, 1.0, -0.5])  # aim inside / outside
        if k % 7 == 0:    # exactly at a corner or along an edge: u, v, u + v on their bounds
            w = np.array([[1, 0, 0], [0, 1, 0], [0.5, 0.5, 0.0], [0, 0.5, 0.5]][(k // 7) % 4], dtype=np.float64)
        target = (w[:, None] * corners.astype(np.float64)).sum(0)
        d = (target - o).astype(f32)
        d = d / f32(np.linalg.norm(d)) * f32(rng.uniform(0.2, 3.0))  # (unnormalised directions too: refracted rays are)
        ray = np.concatenate([o, d]).astype(f32)
        min_dist = 0.001 if k % 5 else 0.5
        ok, t, dist, n = _oracle_tri(oracle, corners, mat, ray, min_dist, 12.0 if k % 4 == 0 else 2000.0)
        got = R.basic_triangle_hit(corners, ray[:3], ray[3:], min_dist, 12.0 if k % 4 == 0 else 2000.0)
        assert ok == (got is not None), k
        if ok:
            n_hit += 1
            assert bits(t) == bits(got[3]) and bits(dist) == bits(got[2]) and np.array_equal(bits(n), bits(got[1])), k
    assert 300 < n_hit < 1300
    # NaN in the ray: every compare is false, `contains` is false -> no hit, and no exception
    ok, *_ = _oracle_tri(oracle, scenes.TRIANGLES[2][0], mat, np.array([0, 1, 0, np.nan, 0, -1], f32))
    assert not ok and R.basic_triangle_hit(np.array(scenes.TRIANGLES[2][0], f32), np.zeros(3, f32), np.array([np.nan, 0, -1], f32), 0.001, 2000.0) is None


def _np_scene(sc):
    mat = lambda m: (int(m.kind), np.array(list(m.albedo), f32), f32(m.param))  # noqa: E731
    meshes = []
    for md in sc.meshes:
        d = {k: md.arrays[k] for k in md.arrays}
        d.update(is_padding=md.is_padding, bbox_lo=np.float32(md.bbox_lo), bbox_hi=np.float32(md.bbox_hi), mat=mat(md.struct.mat))
        meshes.append(d)
    order = None if sc.element_order is None else [("t" if e >> 31 else "s", e & 0x7FFFFFFF) for e in sc.element_order]
    return dict(spheres=[(np.array(c, f32), f32(r), mat(m)) for c, r, m in sc.spheres], meshes=meshes,
                triangles=[(np.array(c, f32), mat(m)) for c, m in sc.triangles], order=order)


def test_whole_pixels_oracle_equals_numpy(oracle):
    """Camera ray -> recursion through spheres, BasicTriangles and a mesh -> per-pixel mean, both restatements."""
    sc = scenes.triangle_scene(oracle, 203)
    W, H, spp, seed = 120, 90, 3, 5
    cam = scenes.camera(oracle, W, H)
    import test_np_reference as T
    nc = T.np_cam(cam)
    ns = _np_scene(sc)
    for (c0, c1, r0, r1) in [(52, 58, 44, 48), (30, 36, 30, 34), (70, 76, 60, 64)]:  # glass triangle / mirror quad / ground triangle
        exp, _, rays = oracle.render(cam, sc, abi.default_opts(spp=spp, seed=seed), window=(c0, c1, r0, r1))
        counter = [0]
        for row in range(r0, r1):
            for col in range(c0, c1):
                got = R.pixel(nc, ns, row, col, spp, seed, counter=counter)
                assert np.array_equal(bits(got), bits(exp[row, col])), (row, col)
        assert counter[0] == rays


@pytest.mark.gpu
def test_scene_hit_through_elements_equals_the_oracle(hip, oracle):
    sc = scenes.triangle_scene(oracle, 1203)
    rng = np.random.default_rng(9)
    n = 4000
    o = np.float32([0.0, 5.0, 4.0]) + rng.normal(size=(n, 3)) * 0.7
    tgt = np.float32([-1.0, 1.5, -9.0]) + rng.uniform(-9, 9, (n, 3)) * np.float32([1.0, 0.5, 1.2])
    d = tgt - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.5, 1.5, (n, 1))
    rays = np.concatenate([o, d], 1).astype(np.float32)
    et, eobj, etri, edist = oracle.trace_rays(sc, rays)
    with hip.HipScene(sc) as hs:
        t, obj, tri, dist = hs.trace_rays(rays)
    assert np.array_equal(obj, eobj) and np.array_equal(tri, etri)
    assert np.array_equal(bits(t), bits(et)) and np.array_equal(bits(dist), bits(edist))
    hit_tri = np.isin(obj, [1, 3, 5, 7])  # the triangles' element ids under scenes.TRIANGLE_ORDER
    assert hit_tri.sum() > 200 and (obj == 8).sum() > 20  # ... and the mesh, object id n_elements + 0


@pytest.mark.gpu
def test_ties_go_to_the_earlier_element(hip, oracle):
    """Two coincident triangles with different materials: scene.rs:27's strict `<` keeps the one tested first. Swapping
    their places in Scene::elements swaps the image; both orders equal the oracle."""
    quad = ((-3.0, 0.2, -8.0), (3.0, 0.2, -8.5), (0.0, 5.0, -8.2))
    red, mirror = abi.material(abi.MAT_LAMBERTIAN, (0.9, 0.1, 0.1)), abi.material(abi.MAT_METAL, (0.9, 0.9, 0.9), 0.0)
    cam = scenes.camera(oracle, 96, 72)
    imgs = []
    for tris in ([(quad, red), (quad, mirror)], [(quad, mirror), (quad, red)]):
        sc = abi.SceneData(spheres=scenes.EXAMPLE_SPHERES[:1], triangles=tris, element_order=[scenes.T | 0, 0, scenes.T | 1])
        got, _ = hip.render_scene(cam, 6, sc, seed=2)
        exp, _, _ = oracle.render(cam, sc, abi.default_opts(spp=6, seed=2))
        assert np.array_equal(bits(got), bits(exp))
        imgs.append(got)
    assert not np.array_equal(imgs[0], imgs[1])


@pytest.mark.gpu
def test_images_with_and_without_an_explicit_order(hip, oracle):
    cam = scenes.camera(oracle, 128, 96)
    for order in (scenes.TRIANGLE_ORDER, None):
        sc = scenes.triangle_scene(oracle, 2004, order=order)
        got, got8 = hip.render_scene(cam, 5, sc, seed=8)
        exp, exp8, _ = oracle.render(cam, sc, abi.default_opts(spp=5, seed=8))
        assert np.array_equal(bits(got), bits(exp)) and np.array_equal(got8, exp8)


@pytest.mark.gpu
@pytest.mark.parametrize("with_mesh", [False, True])
def test_a_permuted_order_of_spheres_alone(hip, oracle, with_mesh):
    """A scene WITHOUT triangle elements whose `element_order` is not the identity (round-3 review: the kernels' sphere-only
    loop takes element e for device sphere e, while the material table follows the order). Two COINCIDENT spheres of different
    materials in front of the example scene: scene.rs:27's strict `<` keeps the element tested first, so order [1, 0, ...]
    must show the mirror where [0, 1, ...] shows the red ball; object ids, materials and images equal the oracle's."""
    red, mirror = abi.material(abi.MAT_LAMBERTIAN, (0.9, 0.1, 0.1)), abi.material(abi.MAT_METAL, (0.9, 0.9, 0.9), 0.0)
    spheres = [((0.5, 1.0, -6.0), 1.0, red), ((0.5, 1.0, -6.0), 1.0, mirror)] + list(scenes.EXAMPLE_SPHERES)
    meshes = [scenes.standin_mesh(oracle, 803, **scenes.EXAMPLE_MESH)] if with_mesh else []
    cam = scenes.camera(oracle, 96, 72)
    rng = np.random.default_rng(4)
    o = np.float32([0.0, 3.0, 4.0]) + rng.normal(size=(2000, 3)).astype(np.float32) * 0.5
    d = np.float32([0.5, 1.0, -6.0]) + rng.uniform(-4, 4, (2000, 3)).astype(np.float32) - o
    rays = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1).astype(np.float32)
    imgs = []
    for order in (None, [1, 0, 2, 3, 4, 5], [5, 4, 1, 3, 0, 2]):
        sc = abi.SceneData(spheres=spheres, meshes=meshes, element_order=order)
        got, _ = hip.render_scene(cam, 6, sc, seed=2)
        exp, _, _ = oracle.render(cam, sc, abi.default_opts(spp=6, seed=2))
        assert np.array_equal(bits(got), bits(exp)), order
        imgs.append(got)
        et, eobj, etri, edist = oracle.trace_rays(sc, rays)
        with hip.HipScene(sc) as hs:
            t, obj, tri, dist = hs.trace_rays(rays)
        assert np.array_equal(obj, eobj) and np.array_equal(bits(t), bits(et)) and np.array_equal(bits(dist), bits(edist)), order
        # object ids are positions in Scene::elements: of the coincident pair, the one tested first takes every hit
        pos = [(order or list(range(6))).index(k) for k in (0, 1)]
        assert (obj == min(pos)).sum() > 50 and (obj == max(pos)).sum() == 0, (order, np.bincount(obj[obj >= 0]))
    # (the second and third order both test the mirror first: the same image; the default order shows the red ball)
    assert not np.array_equal(imgs[0], imgs[1]) and np.array_equal(bits(imgs[1]), bits(imgs[2]))


def test_a_malformed_element_order_is_refused():
    lib = abi.load_hip()
    h = C.c_void_p()
    for order in ([0, 0, 1, 2, scenes.T | 0, scenes.T | 1, scenes.T | 2, scenes.T | 3],   # sphere 0 twice, sphere 3 missing
                  [0, 1, 2, 3, scenes.T | 0, scenes.T | 1, scenes.T | 2, scenes.T | 9]):  # triangle index out of range
        sc = scenes.triangle_scene(order=order)
        assert lib.rbrt_hip_scene_create(sc.ptr(), 0, C.byref(h)) == abi.RBRT_ERR_INVALID_ARG
        assert b"element_order" in lib.rbrt_hip_last_error()
