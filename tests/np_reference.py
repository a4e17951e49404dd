"""A SECOND, independent restatement of the reference's hot path: numpy float32, written from the Rust text.

TEST INFRASTRUCTURE. The C++ oracle (oracle/rbrt_oracle.cpp) is what the GPU is compared with; this file exists
so that the oracle itself is cross-checked by something that shares no code with it, for the rows of SURVEY 8(a)
no reference unit test pins (T1, T4, T5, T6, C1, I1, I2, S0, S1 incl. its quirks, M1-M5). Every function cites the
Rust lines it follows; the arithmetic is numpy float32 scalar/array arithmetic (IEEE, correctly rounded + - * /
sqrt, no FMA), evaluated in the order the Rust expressions evaluate.

The only shared convention is the build's own random stream (DESIGN.md "RNG"; the reference has no seed): it is
restated here with Python integers.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
F0, F1, F2 = f32(0.0), f32(1.0), f32(2.0)
M64 = (1 << 64) - 1


# ---------------------------------------------------------------------------------------------------------------
# random stream (the build's contract, not the reference's): xoroshiro64** keyed by splitmix64
# ---------------------------------------------------------------------------------------------------------------
def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def _rotl32(x: int, k: int) -> int:
    return ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF


class Rng:
    def __init__(self, seed: int, pixel: int, sample: int):
        key = _splitmix64(_splitmix64(seed) ^ ((pixel << 32) | sample))
        self.s0, self.s1 = key & 0xFFFFFFFF, key >> 32
        if self.s0 == 0 and self.s1 == 0:
            self.s0 = 1
        self.draws = 0

    def next_f32(self) -> np.float32:  # rand 0.8 Standard for f32: (u32 >> 8) * 2^-24
        r = (_rotl32((self.s0 * 0x9E3779BB) & 0xFFFFFFFF, 5) * 5) & 0xFFFFFFFF
        t = self.s1 ^ self.s0
        self.s0 = _rotl32(self.s0, 26) ^ t ^ ((t << 9) & 0xFFFFFFFF)
        self.s1 = _rotl32(t, 13)
        self.draws += 1
        return f32(r >> 8) * f32(2.0 ** -24)


# ---------------------------------------------------------------------------------------------------------------
# vec3.rs: Vec3 as a float32 array of shape (3,)
# ---------------------------------------------------------------------------------------------------------------
def vec(x, y, z):
    return np.array([x, y, z], dtype=f32)


def vsum(a):  # vec3.rs:115-117  self.x + self.y + self.z
    return f32(f32(a[0] + a[1]) + a[2])


def dot(a, b):  # vec3.rs:157-159  (*self * *other).sum()
    return vsum(a * b)


def length(a):  # vec3.rs:111-113
    return np.sqrt(f32(f32(a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]))


def normalize(a):  # vec3.rs:119-126: three divisions by the length
    return a / length(a)


def cross(a, b):  # vec3.rs:128-134
    return vec(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def point_at(o, d, t):  # ray.rs:10-12  origin + t * direction
    return o + t * d


# ---------------------------------------------------------------------------------------------------------------
# cam.rs:64-82
# ---------------------------------------------------------------------------------------------------------------
def camera_ray(cam, row: int, col: int, rng: Rng):
    """cam: dict with position, right, up, img_center_point (f32 arrays), mm_per_pix_hor/vert (f32), W, H (ints)."""
    col_off = f32(col) - f32(cam["W"] // 2)  # integer halving first (cam.rs:65)
    row_off = f32(row) - f32(cam["H"] // 2)
    col_mm = f32(f32(col_off + rng.next_f32()) - f32(0.5)) * cam["mm_per_pix_hor"]  # column jitter drawn first
    row_mm = f32(f32(row_off + rng.next_f32()) - f32(0.5)) * cam["mm_per_pix_vert"]
    target = (cam["img_center_point"] + f32(f32(0.001) * col_mm) * cam["right"]) - f32(f32(0.001) * row_mm) * cam["up"]
    return cam["position"].copy(), normalize(target - cam["position"])


# ---------------------------------------------------------------------------------------------------------------
# sphere.rs:20-66
# ---------------------------------------------------------------------------------------------------------------
class NanDiscriminant(Exception):  # the reference panics: partial_cmp(..).expect("Encountered NAN")
    pass


def sphere_hit(center, radius, o, d, min_dist, max_dist):
    """Returns None or (hit_point, hit_normal, dist_from_ray_orig, ray_param)."""
    a = dot(d, d)
    l = o - center
    b = dot(d * F2, l)
    c = dot(l, l) - radius * radius  # powf(2.0)
    sol = b * b - f32(f32(4.0) * a) * c
    if np.isnan(sol):
        raise NanDiscriminant()
    if sol < F0:
        return None
    two = sol > F0
    t = f32(-b - np.sqrt(sol)) / f32(F2 * a)
    if two and t < F0:
        t = f32(-b + np.sqrt(sol)) / f32(F2 * a)
        if t < F0:
            return None
    p = point_at(o, d, t)
    dist = length(o - p)
    if dist < min_dist or dist > max_dist:
        return None
    return p, p - center, dist, t


# ---------------------------------------------------------------------------------------------------------------
# aabbox.rs:28-58 (f32::min / f32::max ignore a NaN operand: np.fmin / np.fmax)
# ---------------------------------------------------------------------------------------------------------------
def bbox_hit(lo, hi, o, d) -> bool:
    with np.errstate(all="ignore"):
        t_lower = (lo - o) / d
        t_upper = (hi - o) / d
    t_min_c = np.fmin(t_lower, t_upper)
    t_min = np.fmax(np.fmax(t_min_c[0], t_min_c[1]), t_min_c[2])
    t_max_c = np.fmax(t_lower, t_upper)
    t_max = np.fmin(np.fmin(t_max_c[0], t_max_c[1]), t_max_c[2])
    if t_max < F0:
        return False
    if t_min > t_max:
        return False
    return True


# ---------------------------------------------------------------------------------------------------------------
# triangle.rs:134-262 (all triangles, 8 per iteration -> here: all at once, elementwise = lane-wise) + 392-410
# ---------------------------------------------------------------------------------------------------------------
def _dot_soa(ax, ay, az, bx, by, bz):  # vec3_avx.rs:10-22: mul, mul, mul, (x + y) + z
    return (ax * bx + ay * by) + az * bz


def _cross_soa(ax, ay, az, bx, by, bz):  # vec3_avx.rs:32-45
    return ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx


def triangle_scan(mesh, o, d, min_dist):
    """mesh: dict of the SoA arrays v0x.. e2z (f32, n_total entries) and is_padding. Returns (t, idx) or None, and
    the per-triangle result array (t or -1000), like ray_params in triangle.rs:142,257-258."""
    n = (len(mesh["v0x"]) // 8) * 8  # chunks_exact(8): the tail is dropped (triangle.rs:166-167)
    eps = f32(min_dist)
    eps_frac = F1 / eps
    g = {k: mesh[k][:n] for k in ("v0x", "v0y", "v0z", "e1x", "e1y", "e1z", "e2x", "e2y", "e2z")}
    with np.errstate(all="ignore"):
        hx, hy, hz = _cross_soa(d[0], d[1], d[2], g["e2x"], g["e2y"], g["e2z"])     # h = d x e_b
        a = _dot_soa(g["e1x"], g["e1y"], g["e1z"], hx, hy, hz)                       # a = e_a . h
        c1 = (-eps < a) & (a < eps)                                                  # ordered compares: False on NaN
        f = F1 / a
        sx, sy, sz = o[0] - g["v0x"], o[1] - g["v0y"], o[2] - g["v0z"]
        u = f * _dot_soa(sx, sy, sz, hx, hy, hz)
        c2 = (u < F0) | (u > F1)
        qx, qy, qz = _cross_soa(sx, sy, sz, g["e1x"], g["e1y"], g["e1z"])            # q = s x e_a
        v = f * _dot_soa(d[0], d[1], d[2], qx, qy, qz)
        c3 = (v < F0) | ((u + v) > F1)
        t = f * _dot_soa(g["e2x"], g["e2y"], g["e2z"], qx, qy, qz)
        c4 = (t > eps) & (t < eps_frac)
    has = ~(c1 | (c2 | c3)) & c4
    params = np.where(has, t, f32(-1000.0)).astype(f32)
    # find_smallest_element_bigger_than_eps, triangle.rs:392-410: strict <, first index wins
    ok = (params > eps) & (params < f32(1000000.0)) & (mesh["is_padding"][:n] == 0)
    if not ok.any():
        return None, params
    cand = np.where(ok, params, f32(np.inf))
    idx = int(np.argmin(cand))  # argmin returns the FIRST minimum
    tmin = cand[idx]
    if tmin > eps and tmin < f32(100000.0):
        return (f32(tmin), idx), params
    return None, params


def mesh_hit(mesh, o, d, min_dist, max_dist):  # mesh.rs:225-267
    if not bbox_hit(mesh["bbox_lo"], mesh["bbox_hi"], o, d):
        return None
    res, _ = triangle_scan(mesh, o, d, min_dist)
    if res is None:
        return None
    t, idx = res
    p = point_at(o, d, t)
    dist = length(o - p)
    if dist > min_dist and dist < max_dist:
        return p, vec(mesh["nx"][idx], mesh["ny"][idx], mesh["nz"][idx]), dist, t, idx
    return None


# ---------------------------------------------------------------------------------------------------------------
# scene.rs:19-43
# ---------------------------------------------------------------------------------------------------------------
def basic_triangle_hit(corners, o, d, min_dist, max_dist):
    """triangle.rs:92-130 + 412-441 (BasicTriangle): returns None or (point, normal, dist, t). corners: (3,3) f32."""
    eps = f32(min_dist)
    v0 = corners[0]
    e0, e1 = corners[1] - corners[0], corners[2] - corners[0]  # triangle.rs:25
    h = cross(d, e1)
    a = dot(e0, h)
    if -eps < a < eps:
        return None
    f = f32(f32(1.0) / a)
    s = o - v0
    u = f32(f * dot(s, h))
    if not (f32(0.0) <= u <= f32(1.0)):  # (0.0..=1.0).contains(&u): false for NaN
        return None
    q = cross(s, e0)
    v = f32(f * dot(d, q))
    if v < f32(0.0) or f32(u + v) > f32(1.0):
        return None
    t = f32(f * dot(e1, q))
    if not (t > eps):
        return None
    p = point_at(o, d, t)
    dist = length(o - p)
    if dist < f32(min_dist) or dist > f32(max_dist):
        return None
    return p, normalize(cross(e0, e1)), dist, t  # triangle.rs:30-34: stored at construction, never flipped


def scene_hit(scene, o, d, min_dist, max_dist):
    """scene: dict(spheres=[(center, radius, mat)], meshes=[mesh dicts with 'mat'], optional triangles=[(corners, mat)] and
    order=[(kind, index)] = Scene::elements order, kind 's' or 't'). Returns None or a dict."""
    best, closest = None, np.finfo(f32).max
    tris = scene.get("triangles", [])
    order = scene.get("order") or [("s", i) for i in range(len(scene["spheres"]))] + [("t", i) for i in range(len(tris))]
    for e, (kind, i) in enumerate(order):
        if kind == "s":
            c, r, m = scene["spheres"][i]
            h = sphere_hit(c, r, o, d, min_dist, max_dist)
        else:
            corners, m = tris[i]
            h = basic_triangle_hit(corners, o, d, min_dist, max_dist)
        if h is not None and h[2] < closest:
            closest = h[2]
            best = dict(point=h[0], normal=h[1], dist=h[2], t=h[3], mat=m, obj=e, tri=-1)
    for k, mesh in enumerate(scene["meshes"]):
        h = mesh_hit(mesh, o, d, min_dist, max_dist)
        if h is not None and h[2] < closest:
            closest = h[2]
            best = dict(point=h[0], normal=h[1], dist=h[2], t=h[3], mat=mesh["mat"], obj=len(order) + k, tri=h[4])
    return best


# ---------------------------------------------------------------------------------------------------------------
# materials.rs:14-37, lambertian.rs:11-24, metal.rs:12-25, dielectric.rs:11-85
# ---------------------------------------------------------------------------------------------------------------
def random_point_in_unit_sphere(rng: Rng):
    one = vec(1, 1, 1)
    p = F2 * vec(rng.next_f32(), rng.next_f32(), rng.next_f32()) - one  # x, y, z drawn in that order
    while length(p) > F1:
        p = F2 * vec(rng.next_f32(), rng.next_f32(), rng.next_f32()) - one
    return p


def reflect(direction, normal):
    du = normalize(direction)
    nu = normalize(normal)
    return normalize(du - (F2 * nu) * dot(du, nu))


def schlick(cosine, ref_index):  # powi(2) = x*x, powi(5) = x * (x^2)^2 (LLVM's multiply chain)
    q = f32(F1 - ref_index) / f32(F1 + ref_index)
    r0 = q * q
    x = F1 - cosine
    x2 = x * x
    return r0 + f32(F1 - r0) * f32(x * f32(x2 * x2))


def refract(direction, normal, ni_over_nt):
    vu = normalize(direction)
    nu = normalize(normal)
    cos_theta = dot(vu, nu)
    discr = F1 - f32(ni_over_nt * ni_over_nt) * f32(F1 - cos_theta * cos_theta)
    if discr > F0:
        return ni_over_nt * (vu - nu * cos_theta) - np.sqrt(discr) * nu  # not re-normalised
    return None


KIND_LAMBERTIAN, KIND_METAL, KIND_DIELECTRIC = 0, 1, 2


def scatter(mat, d_in, hit, rng: Rng):
    """mat: (kind, albedo f32[3], param f32). Returns (ok, attenuation, new_origin, new_direction)."""
    kind, albedo, param = mat
    p, n = hit["point"], hit["normal"]
    if kind == KIND_LAMBERTIAN:
        target = (p + normalize(n)) + random_point_in_unit_sphere(rng)
        return True, albedo, p, normalize(target - p)
    if kind == KIND_METAL:
        target = reflect(d_in, n)
        nd = normalize(target + param * random_point_in_unit_sphere(rng))
        return bool(dot(nd, n) > F0), albedo, p, nd
    reflected = reflect(d_in, n)
    a = dot(normalize(d_in), normalize(n))
    if a > F0:
        outward, ni_over_nt, cosine = f32(-1.0) * n, param, param * a
    else:
        outward, ni_over_nt, cosine = n, F1 / param, -a
    refracted = refract(d_in, outward, ni_over_nt)
    reflect_prob = schlick(cosine, param) if refracted is not None else F1
    if refracted is None:
        refracted = vec(0, 0, 0)
    u = rng.next_f32()  # always exactly one draw (dielectric.rs:48)
    return True, vec(1, 1, 1), p, (reflected if u < reflect_prob else refracted)


# ---------------------------------------------------------------------------------------------------------------
# lib.rs:43-73 (colorize), 84-113 (pixel loop), 116-122 (quantise)
# ---------------------------------------------------------------------------------------------------------------
def colorize(o, d, scene, bg, depth: int, rng: Rng, counter=None):
    if counter is not None:
        counter[0] += 1
    hit = scene_hit(scene, o, d, f32(0.001), f32(2000.0))
    if hit is not None:
        if depth > 0:  # `current_depth > 0 && scatter(..)`: scatter (and its draws) only when depth > 0
            ok, att, no, nd = scatter(hit["mat"], d, hit, rng)
            if ok:
                return att * colorize(no, nd, scene, bg, depth - 1, rng, counter)  # attenuation * colorize(..): right to left
        return vec(0, 0, 0)
    t = f32(0.5) * f32(d[1] + F1)
    return t * vec(1, 1, 1) + f32(F1 - t) * bg


def pixel(cam, scene, row: int, col: int, spp: int, seed: int, max_depth: int = 50, counter=None):
    bg = vec(0.05, 0.05, 0.8)
    color = vec(0, 0, 0)
    for s in range(spp):
        rng = Rng(seed, row * cam["W"] + col, s)
        o, d = camera_ray(cam, row, col, rng)
        color = color + colorize(o, d, scene, bg, max_depth, rng, counter)
    return color * f32(F1 / f32(spp))


def quantise(c):  # (c.sqrt() * 256.0) as u8: saturating, NaN -> 0
    v = np.sqrt(f32(c)) * f32(256.0)
    if np.isnan(v) or v <= 0:
        return 0
    return 255 if v >= 255 else int(v)
