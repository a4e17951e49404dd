"""Pins the CPU oracle against every known-answer test the reference's own unit tests hold for
the hot path. Each test names the reference test it restates. All exact
(assert_eq! in the reference) unless the reference itself uses a tolerance.
"""
import ctypes as C

import numpy as np
import pytest

from rbrt_amd import abi

f32 = np.float32


def v(*a):
    return np.array(a, dtype=np.float32)


def vec3(oracle, op, a, b=(0, 0, 0)):
    out = np.zeros(3, f32)
    oracle.lib().rbrt_oracle_kat_vec3(op, oracle._p(v(*a)), oracle._p(v(*b)), oracle._p(out))
    return out


CROSS, NORMALIZE, DOT, MUL, ADD, SUB, ROTATE, LENGTH, REFLECT = range(9)


def test_vec3_cross(oracle):  # vec3.rs:170-175 test_cross_product
    assert np.array_equal(vec3(oracle, CROSS, (1, 0, 0), (0, 1, 0)), v(0, 0, 1))


def test_vec3_normalize_length_is_exactly_one(oracle):  # vec3.rs:177-179 test_normalize
    n = vec3(oracle, NORMALIZE, (5, 2, 3))
    assert vec3(oracle, LENGTH, n)[0] == f32(1.0)


def test_vec3_dot(oracle):  # vec3.rs:181-186 test_dot_product
    assert vec3(oracle, DOT, (1, 2, 3), (1, 2, 3))[0] == f32(14.0)


def test_vec3_mul_add_sub(oracle):  # vec3.rs:188-208 test_mul, test_add, test_subtract
    assert np.array_equal(vec3(oracle, MUL, (1, 2, 3), (1, 2, 3)), v(1, 4, 9))
    assert np.array_equal(vec3(oracle, ADD, (1, 2, 3), (1, 2, 3)), v(2, 4, 6))
    assert np.array_equal(vec3(oracle, SUB, (1, 2, 3), (1, 2, 3)), v(0, 0, 0))


A = 0.7071067657322372
R45 = float(np.float32(np.deg2rad(np.float32(45.0))))
# vec3.rs:226-325: (point, rotation, expected), 19 rows, tolerance 1e-6 on the residual length
ROT_TABLE = [
    ((1, 0, 0), (0, 0, R45), (A, A, 0)), ((1, 0, 0), (0, 0, -R45), (A, -A, 0)),
    ((0, 1, 0), (0, 0, R45), (-A, A, 0)), ((0, 1, 0), (0, 0, -R45), (A, A, 0)),
    ((0, 0, 1), (0, 0, R45), (0, 0, 1)), ((0, 0, 1), (0, 0, -R45), (0, 0, 1)),
    ((1, 0, 0), (0, R45, 0), (1, 0, 0)), ((1, 0, 0), (0, -R45, 0), (1, 0, 0)),
    ((0, 1, 0), (0, R45, 0), (0, A, A)), ((0, 1, 0), (0, -R45, 0), (0, A, -A)),
    ((0, 0, 1), (0, R45, 0), (0, -A, A)), ((0, 0, 1), (0, -R45, 0), (0, A, A)),
    ((1, 0, 0), (R45, 0, 0), (A, A, 0)), ((1, 0, 0), (-R45, 0, 0), (A, -A, 0)),
    ((0, 1, 0), (R45, 0, 0), (-A, A, 0)), ((0, 1, 0), (-R45, 0, 0), (A, A, 0)),
    ((0, 0, 1), (R45, 0, 0), (0, 0, 1)), ((0, 0, 1), (R45, 0, 0), (0, 0, 1)),
    ((0, 0, 1), (-R45, 0, 0), (0, 0, 1)),
]


@pytest.mark.parametrize("pt,rot,exp", ROT_TABLE)
def test_vec3_rotate_table(oracle, pt, rot, exp):  # vec3.rs:209-342 test_rotate_yaw
    got = vec3(oracle, ROTATE, pt, rot)
    residuum = np.linalg.norm(v(*exp).astype(np.float64) - got.astype(np.float64))
    assert residuum < 1e-6


# vec3_avx.rs:59-110 (the SSE tests vec3_sse.rs:58-172 use the same numbers 4-wide)
AX = [1.0, 0.0, 3.0, 2.0, 1.0, 0.0, 3.0, 2.0]
AY = [0.0, 1.0, 4.0, 6.0, 0.0, 1.0, 4.0, 6.0]
AZ = [0.0, 0.0, 4.0, 3.0, 0.0, 0.0, 4.0, 3.0]
BX = [0.0, 0.0, 1.0, 2.0, 0.0, 0.0, 1.0, 2.0]
BY = [1.0, 0.0, -2.0, 1.0, 1.0, 0.0, -2.0, 1.0]
BZ = [0.0, 1.0, 3.0, -2.0, 0.0, 1.0, 3.0, -2.0]


def test_avx_cross_product(oracle):  # vec3_avx.rs:59-87 test_avx_cross_product
    a, b = v(*(AX + AY + AZ)), v(*(BX + BY + BZ))
    out = np.zeros(24, f32)
    oracle.lib().rbrt_oracle_kat_avx(0, oracle._p(a), oracle._p(b), oracle._p(out))
    assert np.array_equal(out[0:8], v(0, 1, 20, -15, 0, 1, 20, -15))
    assert np.array_equal(out[8:16], v(0, 0, -5, 10, 0, 0, -5, 10))
    assert np.array_equal(out[16:24], v(1, 0, -10, -10, 1, 0, -10, -10))


def test_avx_dot_product(oracle):  # vec3_avx.rs:89-110 test_avx_dot_product
    a, b = v(*(AX + AY + AZ)), v(*(BX + BY + BZ))
    out = np.zeros(8, f32)
    oracle.lib().rbrt_oracle_kat_avx(1, oracle._p(a), oracle._p(b), oracle._p(out))
    assert np.array_equal(out, v(0, 0, 7, 4, 0, 0, 7, 4))


def test_triangle_normal(oracle):  # triangle.rs:448-475 test_triangle_normal
    out = np.zeros(3, f32)
    oracle.lib().rbrt_oracle_kat_triangle_normal(oracle._p(v(1, 0, 0, 1, 1, 0, 0, 0, 0)), oracle._p(out))
    assert np.array_equal(out, v(0, 0, 1))
    oracle.lib().rbrt_oracle_kat_triangle_normal(oracle._p(v(1, 0, 0, 1, 0, 1, 0, 1, 0)), oracle._p(out))
    assert np.array_equal(out, vec3(oracle, NORMALIZE, (-1, -1, 0)))


def test_sphere_intersection(oracle):  # sphere.rs:75-112 test_sphere_intersection
    s = abi.Sphere((C.c_float * 3)(0, 0, -10), 1.0, abi.material(abi.MAT_METAL, (0.8, 0.8, 0.8), 0.005))
    p, n, d = np.zeros(3, f32), np.zeros(3, f32), C.c_float()
    ok = oracle.lib().rbrt_oracle_kat_sphere(C.byref(s), oracle._p(v(0, 0, 0, 0, 0, -1)), 0.001, 1000.0,
                                             oracle._p(p), oracle._p(n), C.byref(d))
    assert ok == 1
    assert np.array_equal(p, v(0, 0, -9)) and np.array_equal(n, v(0, 0, 1))
    ok = oracle.lib().rbrt_oracle_kat_sphere(C.byref(s), oracle._p(v(0, 0, -15, 0, 0, 1)), 0.001, 1000.0,
                                             oracle._p(p), oracle._p(n), C.byref(d))
    assert ok == 1
    assert np.array_equal(p, v(0, 0, -11)) and np.array_equal(n, v(0, 0, -1))


def test_random_points_in_unit_sphere(oracle):  # materials.rs:43-47 (20 draws, length < 1)
    out = np.zeros(3 * 20, f32)
    oracle.lib().rbrt_oracle_kat_unit_sphere(1, 0, 0, 20, oracle._p(out))
    lens = np.sqrt((out.reshape(20, 3).astype(np.float64) ** 2).sum(1))
    assert (lens < 1.0).all()


def test_reflection(oracle):  # materials.rs:49-59 test_reflection
    refl = vec3(oracle, REFLECT, (1, 1, 1), (1, 1, 1))
    assert np.array_equal(refl, f32(-1.0) * vec3(oracle, NORMALIZE, (1, 1, 1)))
    refl = vec3(oracle, REFLECT, (1, 1, 0), (-1, 0, 0))
    assert np.array_equal(refl, v(-0.7071068, 0.7071068, 0.0))


def test_refraction(oracle):  # dielectric.rs:93-115 test_refraction
    inc = vec3(oracle, NORMALIZE, (1, 1, 0))
    nrm = vec3(oracle, NORMALIZE, (-1, 0, 0))
    out = np.zeros(3, f32)
    ok = oracle.lib().rbrt_oracle_kat_refract(oracle._p(inc), oracle._p(nrm), 1.4, oracle._p(out))
    assert ok == 1
    assert np.array_equal(out, v(0.14142191, 0.9899495, 0.0))


def test_mesh_aabbox(oracle):  # aabbox.rs:95-108 test_mesh_aabbox
    md = oracle.mesh_prep(np.array([[[1, 0, 0], [1, 0, 1], [0, 1, 0]]], np.float32))
    assert np.array_equal(md.bbox_lo, v(0, 0, 0)) and np.array_equal(md.bbox_hi, v(1, 1, 1))


# ---- things no reference test pins: checked against an independent numpy-f32 restatement ------

def test_rng_mapping_and_determinism(oracle):
    raw = np.zeros(64, np.uint32)
    flt = np.zeros(64, f32)
    oracle.lib().rbrt_oracle_kat_rng(1, 7, 3, 64, raw.ctypes.data_as(C.POINTER(C.c_uint32)), oracle._p(flt))
    # rand 0.8 Standard for f32: top 24 bits * 2^-24
    assert np.array_equal(flt, (raw >> 8).astype(f32) * f32(2.0 ** -24))
    assert (flt >= 0).all() and (flt < 1).all()
    raw2 = np.zeros(64, np.uint32)
    oracle.lib().rbrt_oracle_kat_rng(1, 7, 4, 64, raw2.ctypes.data_as(C.POINTER(C.c_uint32)), None)
    assert not np.array_equal(raw, raw2)
    # independent restatement of the stream (python ints)
    M64 = (1 << 64) - 1

    def sm(x):
        x = (x + 0x9E3779B97F4A7C15) & M64
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def rotl(x, k):
        return ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF

    key = sm(sm(1) ^ ((7 << 32) | 3))
    s0, s1 = key & 0xFFFFFFFF, key >> 32
    exp = []
    for _ in range(64):
        exp.append((rotl((s0 * 0x9E3779BB) & 0xFFFFFFFF, 5) * 5) & 0xFFFFFFFF)
        t = s1 ^ s0
        s0 = rotl(s0, 26) ^ t ^ ((t << 9) & 0xFFFFFFFF)
        s1 = rotl(t, 13)
    assert raw.tolist() == exp


def test_padding_rule(oracle):  # mesh.rs:134-144: pads N % 8 copies of triangle 0
    rng = np.random.default_rng(0)
    for n in (1, 3, 8, 11, 12, 14, 16):
        soup = rng.uniform(-1, 1, (n, 3, 3)).astype(f32)
        md = oracle.mesh_prep(soup)
        assert md.n_total == n + n % 8
        assert md.is_padding[:n].sum() == 0 and md.is_padding[n:].all()
        for k in ("v0x", "e1y", "nz"):
            assert (md.arrays[k][n:] == md.arrays[k][0]).all()


def test_truncation_quirk(oracle):
    """chunks_exact(8) (triangle.rs:166-167): with N % 8 in {1,2,3} the last N % 8 real triangles are
    never tested; with {0,4..7} all are."""
    tri = np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], f32)  # faces +z, hit by a ray down -z
    ray = v(0.2, 0.2, 0, 0, 0, -1)
    for n in range(1, 18):
        soup = np.tile(np.array([[100, 100, 100], [101, 100, 100], [100, 101, 100]], f32), (n, 1, 1))
        soup[n - 1] = tri  # only the LAST triangle is in the ray's way
        md = oracle.mesh_prep(soup)
        t, idx = C.c_float(), C.c_int32()
        ok = oracle.lib().rbrt_oracle_kat_mesh_intersect(C.byref(md.struct), oracle._p(ray), 0.001,
                                                         C.byref(t), C.byref(idx), None)
        visible = (n % 8) not in (1, 2, 3)
        assert bool(ok) == visible, n
        if visible:
            assert idx.value == n - 1 and t.value == 5.0


def test_quantise(oracle):  # lib.rs:116-122, Rust `as u8` saturation
    q = oracle.lib().rbrt_oracle_kat_quantise
    assert q(0.0) == 0 and q(1.0) == 255 and q(4.0) == 255 and q(-1.0) == 0 and q(float("nan")) == 0
    assert q(0.25) == 128  # sqrt(.25)*256
    assert q(0.99) == int(np.sqrt(f32(0.99)) * f32(256.0))


def test_schlick_numpy(oracle):
    for cos, n in ((0.3, 1.8), (0.9, 0.2), (0.0, 1.5)):
        cos, n = f32(cos), f32(n)
        r0 = ((f32(1) - n) / (f32(1) + n)) ** 2
        r0 = f32(((f32(1) - n) / (f32(1) + n)) * ((f32(1) - n) / (f32(1) + n)))
        x = f32(1) - cos
        x2 = f32(x * x)
        x4 = f32(x2 * x2)
        exp = f32(r0 + f32(f32(f32(1) - r0) * f32(x * x4)))
        assert oracle.lib().rbrt_oracle_kat_schlick(float(cos), float(n)) == exp


def test_unit_sphere_rejection_threshold_equivalence():
    """The kernel tests s > nextafter(1) instead of sqrt(s) > 1 (materials.rs:21): identical for every float."""
    one = f32(1.0)
    nxt = np.nextafter(one, f32(2))
    lo, hi = f32(0.99).view(np.uint32), f32(1.01).view(np.uint32)
    s = np.arange(lo, hi + 1, dtype=np.uint32).view(np.float32)
    assert np.array_equal(np.sqrt(s) > one, s > nxt)
    r = np.random.default_rng(0).uniform(0, 3.1, 2_000_000).astype(np.float32)
    assert np.array_equal(np.sqrt(r) > one, r > nxt)
