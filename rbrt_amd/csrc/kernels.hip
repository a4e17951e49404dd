// kernels.hip — gfx950 (MI355X / CDNA4) kernels of the rbrt path-tracing hot path.
//
// What runs here is the whole of the reference's L1 layer for one (pixel, sample):
//   camera ray        cam.rs:64-82
//   closest hit       scene.rs:19-43 -> sphere.rs:20-66, mesh.rs:225-267 (gate aabbox.rs:28-58,
//                     Moller-Trumbore triangle.rs:189-255, arg-min triangle.rs:392-410)
//   scatter           lambertian.rs:11-24, metal.rs:12-25, dielectric.rs:11-85, materials.rs:14-37
//   integrator        lib.rs:43-73 (depth 50, sky gradient), per-pixel mean lib.rs:95-101,
//                     quantisation lib.rs:116-122
// Arithmetic follows the reference operation by operation (f32, mul-then-add, no FMA on any value
// that reaches the image: this file is compiled with -ffp-contract=off, and IEEE-correct / and sqrt)
// so that the radiance matches the CPU restatement of the reference bit for bit. The only place
// the GPU does something the reference does not is culling: triangles are found through a BVH
// instead of a scan over all of them, with node boxes inflated by a per-ray error bound so that no
// triangle the scan would accept can be skipped (mesh_closest below).
//
// Wave64 only; no MFMA (there is no dense contraction on this path).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

#pragma clang fp contract(off)

namespace rbrt {

// ---------------------------------------------------------------------------------------------
// Vec3 with the reference's evaluation order (vec3.rs:12-160)
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 mk(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 b) { return V3{s * b.x, s * b.y, s * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// ---- IEEE square root and division, the short way ----------------------------------------------------------
// The image needs correctly rounded sqrt and / (the reference's f32::sqrt and `/`). hipcc's expansions are 17 and
// 13 instructions: v_sqrt / v_rcp plus correction steps, wrapped in range handling (input scaling for tiny
// operands, v_div_scale / v_div_fixup, class checks for 0 and inf). For operands in the everyday range that
// wrapping is the identity, and the correction steps -- reproduced below instruction for instruction -- give the same
// bits. RBRT_FAST_IEEE: a wave takes the short forms when EVERY active lane's operands are in range (one compare or
// two per call) and the compiler's forms otherwise. tests: test_short_ieee_forms_match_the_compilers (2^28 operands).
#ifndef RBRT_FAST_IEEE
#define RBRT_FAST_IEEE 1
#endif
__device__ __forceinline__ float sqrt_core(float x) {  // x in [2^-80, 2^100]: hipcc's sqrt without its scaling / class steps
    const float r = __builtin_amdgcn_sqrtf(x);
    const float r_dn = __uint_as_float(__float_as_uint(r) - 1u), r_up = __uint_as_float(__float_as_uint(r) + 1u);
    const float e_dn = __builtin_fmaf(-r_dn, r, x), e_up = __builtin_fmaf(-r_up, r, x);
    float s = e_dn <= 0.0f ? r_dn : r;
    s = e_up > 0.0f ? r_up : s;
    return s;
}
__device__ __forceinline__ bool sqrt_in_range(float x) { return x > 0x1p-80f && x < 0x1p100f; }  // (false for NaN)
__device__ __forceinline__ float ieee_sqrt(float x) {
    if (RBRT_FAST_IEEE && __builtin_amdgcn_ballot_w64(!sqrt_in_range(x)) == 0ull) return sqrt_core(x);
    return __builtin_sqrtf(x);
}
__device__ __forceinline__ float length(V3 a) { return ieee_sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ __attribute__((noinline)) V3 normalize_ieee(V3 a) {  // the compiler's forms, out of line: the rare path
    const float len = __builtin_sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
    return V3{a.x / len, a.y / len, a.z / len};
}
__device__ __forceinline__ V3 normalize(V3 a) {  // three divisions by the length, vec3.rs:119-126
    if (!RBRT_FAST_IEEE) {
        const float len = __builtin_sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
        return V3{a.x / len, a.y / len, a.z / len};
    }
    const float s = (a.x * a.x + a.y * a.y) + a.z * a.z;
    const float mn = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    // in range: no scaling in sqrt; len in [2^-40, 2^50]; every |component| >= 2^-100 (so v_div_scale leaves numerator
    // and denominator alone: exponent(num) > 23, |num/len| <= 1 is normal) and <= len; nothing is 0, inf or NaN
    const bool ok = sqrt_in_range(s) && mn > 0x1p-100f;
    if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return normalize_ieee(a);
    const float len = sqrt_core(s);
    // hipcc's fdiv, minus v_div_scale / v_div_fixup, the reciprocal refinement shared by the three numerators
    float rc = __builtin_amdgcn_rcpf(len);
    rc = __builtin_fmaf(__builtin_fmaf(-len, rc, 1.0f), rc, rc);
    V3 q;
    {
        float v = a.x * rc;
        v = __builtin_fmaf(__builtin_fmaf(-len, v, a.x), rc, v);
        q.x = __builtin_fmaf(__builtin_fmaf(-len, v, a.x), rc, v);
        v = a.y * rc;
        v = __builtin_fmaf(__builtin_fmaf(-len, v, a.y), rc, v);
        q.y = __builtin_fmaf(__builtin_fmaf(-len, v, a.y), rc, v);
        v = a.z * rc;
        v = __builtin_fmaf(__builtin_fmaf(-len, v, a.z), rc, v);
        q.z = __builtin_fmaf(__builtin_fmaf(-len, v, a.z), rc, v);
    }
    return q;
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// ---------------------------------------------------------------------------------------------
// Random stream: xoroshiro64** keyed by (seed, pixel, sample); see DESIGN.md "RNG".
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) {
    return __builtin_rotateleft32(x, k);
}
struct Rng {
    uint32_t s0, s1;
    __device__ __forceinline__ void init(uint64_t seed_key, uint32_t pixel, uint32_t sample) {
        uint64_t key = splitmix64(seed_key ^ ((uint64_t(pixel) << 32) | uint64_t(sample)));
        s0 = uint32_t(key);
        s1 = uint32_t(key >> 32);
        if ((s0 | s1) == 0) s0 = 1;
    }
    __device__ __forceinline__ uint32_t next_u32() {
        uint32_t r = rotl32(s0 * 0x9E3779BBu, 5) * 5u;
        uint32_t t = s1 ^ s0;
        s0 = rotl32(s0, 26) ^ t ^ (t << 9);
        s1 = rotl32(t, 13);
        return r;
    }
    // 24-bit uniform in [0,1): rand 0.8's Standard distribution for f32
    __device__ __forceinline__ float next_f32() {
        return float(next_u32() >> 8) * (1.0f / 16777216.0f);
    }
};

// ---------------------------------------------------------------------------------------------
// Per-thread counters (counting variant only)
// ---------------------------------------------------------------------------------------------
struct LocalCounters {
    uint32_t rays, gate, nodes, tris, mesh_hits;
};

// ---------------------------------------------------------------------------------------------
// Geometry
// ---------------------------------------------------------------------------------------------

// sphere.rs:20-66. Returns true on hit with the ray parameter and the distance.
__device__ __forceinline__ bool sphere_hit(V3 c, float radius, V3 o, V3 d, float min_dist,
                                           float max_dist, float& t_out, float& dist_out,
                                           DevCounters* counters) {
    float a = dot(d, d);
    V3 l = o - c;
    float b = dot(d * 2.0f, l);
    float cc = dot(l, l) - radius * radius;
    float sol = b * b - 4.0f * a * cc;
    if (sol != sol) {  // the reference panics here (sphere.rs:33); count it and miss
        atomicAdd(&counters->nan_discriminants, 1ull);
        return false;
    }
    if (sol < 0.0f) return false;
    float t = (-b - ieee_sqrt(sol)) / (2.0f * a);
    if (sol > 0.0f && t < 0.0f) {
        t = (-b + ieee_sqrt(sol)) / (2.0f * a);
        if (t < 0.0f) return false;
    }
    V3 p = o + t * d;
    float dist = length(o - p);
    if (dist < min_dist || dist > max_dist) return false;
    t_out = t;
    dist_out = dist;
    return true;
}

// triangle.rs:92-130 (basic_triangle_intersect_w_ray) + :412-441: one BasicTriangle element. Scalar Moller-Trumbore
// with the reference's tests in the reference's order; `(0.0..=1.0).contains(&u)` is false for a NaN u, the other
// compares are false for NaN operands, so a NaN t ends in `t > eps` being false. No upper bound on t but the
// distance window.
__device__ __forceinline__ bool basic_triangle_hit(const float* tri /* v0, e0, e1 */, V3 o, V3 d, float min_dist, float max_dist,
                                                   float& t_out, float& dist_out) {
    const float eps = min_dist;
    const V3 v0 = mk(tri), e0 = mk(tri + 3), e1 = mk(tri + 6);
    const V3 h = cross(d, e1);
    const float a = dot(e0, h);
    if (-eps < a && a < eps) return false;
    const float f = 1.0f / a;
    const V3 s = o - v0;
    const float u = f * dot(s, h);
    if (!(0.0f <= u && u <= 1.0f)) return false;
    const V3 q = cross(s, e0);
    const float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = f * dot(e1, q);
    if (!(t > eps)) return false;
    const V3 p = o + t * d;
    const float dist = length(o - p);
    if (dist < min_dist || dist > max_dist) return false;
    t_out = t;
    dist_out = dist;
    return true;
}

// aabbox.rs:28-58, verbatim: six true divisions, NaN-ignoring min/max (fminf/fmaxf = f32::min/max).
__device__ __forceinline__ bool bbox_gate(const float* lo, const float* hi, V3 o, V3 d) {
    float t_lower_x = (lo[0] - o.x) / d.x;
    float t_upper_x = (hi[0] - o.x) / d.x;
    float t_lower_y = (lo[1] - o.y) / d.y;
    float t_upper_y = (hi[1] - o.y) / d.y;
    float t_lower_z = (lo[2] - o.z) / d.z;
    float t_upper_z = (hi[2] - o.z) / d.z;
    float t_min_x = __builtin_fminf(t_lower_x, t_upper_x);
    float t_min_y = __builtin_fminf(t_lower_y, t_upper_y);
    float t_min_z = __builtin_fminf(t_lower_z, t_upper_z);
    float t_min = __builtin_fmaxf(__builtin_fmaxf(t_min_x, t_min_y), t_min_z);
    float t_max_x = __builtin_fmaxf(t_lower_x, t_upper_x);
    float t_max_y = __builtin_fmaxf(t_lower_y, t_upper_y);
    float t_max_z = __builtin_fmaxf(t_lower_z, t_upper_z);
    float t_max = __builtin_fminf(__builtin_fminf(t_max_x, t_max_y), t_max_z);
    if (t_max < 0.0f) return false;
    if (t_min > t_max) return false;
    return true;
}

// The same DECISION as bbox_gate for every input, without the six IEEE divisions on the common path: the quotients
// are formed as (bound - o) * v_rcp_f32(d) (1 ulp reciprocal), each within 2 * 2^-23 relative of the correctly
// rounded quotient bbox_gate computes (the subtraction is the same instruction in both). min / max are monotone, so
// t_min and t_max are within E = 4 * 2^-23 * (largest of the six magnitudes) of bbox_gate's values, and both of its
// tests -- t_max < 0, t_min > t_max -- come out the same whenever |t_max| > E and |t_min - t_max| > 2E. Otherwise
// (a ray grazing an edge of the box within a few ulp, a zero / tiny / huge direction component, non-finite
// quotients) the lane falls back to the IEEE form; the wave takes that branch only when one of its lanes needs it.
// `beyond`: a ray parameter past which no triangle of this mesh could be accepted any more (the ray's closest hit so
// far, relaxed as in the megakernel's refill; +inf when there is none). A box entered only past it is skipped: that is
// not a decision of the reference's gate, it is the traversal's first culling step done here, before the ray is parked.
__device__ __forceinline__ bool bbox_gate_fast(const float* lo, const float* hi, V3 o, V3 d, float beyond = __builtin_inff()) {
    const float rx = __builtin_amdgcn_rcpf(d.x), ry = __builtin_amdgcn_rcpf(d.y), rz = __builtin_amdgcn_rcpf(d.z);
    const float lx = (lo[0] - o.x) * rx, ux = (hi[0] - o.x) * rx;
    const float ly = (lo[1] - o.y) * ry, uy = (hi[1] - o.y) * ry;
    const float lz = (lo[2] - o.z) * rz, uz = (hi[2] - o.z) * rz;
    const float t_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(lx, ux), __builtin_fminf(ly, uy)), __builtin_fminf(lz, uz));
    const float t_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(lx, ux), __builtin_fmaxf(ly, uy)), __builtin_fmaxf(lz, uz));
    const float m = __builtin_fmaxf(
        __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(lx), __builtin_fabsf(ux)), __builtin_fmaxf(__builtin_fabsf(ly), __builtin_fabsf(uy))),
        __builtin_fmaxf(__builtin_fabsf(lz), __builtin_fabsf(uz)));
    const float E = m * (4.0f / 8388608.0f);
    const float dmin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    const float dmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    // every comparison below is false for a NaN operand: anything unusual lands in the exact branch
    const bool sure = (__builtin_fabsf(t_max) > E) && (__builtin_fabsf(t_min - t_max) > 2.0f * E) && (m < 1e30f) && (m > 1e-30f) &&
                      (dmin > 1e-30f) && (dmax < 1e30f) && (d.x == d.x) && (d.y == d.y) && (d.z == d.z);
    bool res = !(t_max < 0.0f) && !(t_min > t_max) && !(t_min - E > beyond);
    if (!sure) res = bbox_gate(lo, hi, o, d);
    return res;
}

// One lane of triangle.rs:189-255. Every compare is an ordered compare (false on NaN) and the
// accept expression has the reference's shape: !(c1 | c2 | c3) & c4.
__device__ __forceinline__ bool tri_test(V3 v0, V3 ea, V3 eb, V3 o, V3 d, float eps, float eps_frac,
                                         float& t_out) {
    V3 h = cross(d, eb);
    float a = dot(ea, h);
    bool c1 = (-eps < a) && (a < eps);
    float f = 1.0f / a;
    V3 s = o - v0;
    float u = f * dot(s, h);
    bool c2 = (u < 0.0f) || (u > 1.0f);
    V3 q = cross(s, ea);
    float v = f * dot(d, q);
    bool c3 = (v < 0.0f) || ((u + v) > 1.0f);
    float t = f * dot(eb, q);
    bool c4 = (t > eps) && (t < eps_frac);
    t_out = t;
    return !(c1 || c2 || c3) && c4;
}

// ---------------------------------------------------------------------------------------------
// BVH traversal pieces shared by the megakernel and the simple per-thread path.
//
// None of this arithmetic reaches the image: it only decides which triangles get tested. So it may
// use FMA and the fast reciprocal / sqrt. Why culling cannot change the result: a triangle accepted by
// tri_test with parameter t has its point o + t*d within
//     err <= ~18 * 2^-24 * (K*|d| + 1) * S        K = |e1||e2| / eps,  S >= |o - v0|
// of the triangle's exact surface (|a| >= eps bounds the amplification of the rounding errors in
// u, v and t). Child boxes are grown by pad = 64 * 2^-24 * (K_subtree*|d| + 1) * (S + |o|_inf), so
// that point is strictly inside every ancestor's grown box (the |o|_inf term covers the o*inv
// product of the FMA form), the slab interval contains t, and `tn <= best_t` (not <) keeps
// equal-t candidates with a lower index reachable.
// ---------------------------------------------------------------------------------------------
// n / d for a launch-constant divisor: m = floor(2^32 / d) (host, div_magic_of) under-estimates the quotient by
// at most 2, fixed by two conditional steps; ~8 instructions instead of the ~40 of a 32-bit division.
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t d, uint32_t m) {
    uint32_t q = __umulhi(n, m);
    uint32_t r = n - q * d;
    if (r >= d) ++q, r -= d;
    if (r >= d) ++q;
    return q;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define RBRT_AS1(T, p) ((const __attribute__((address_space(1))) T*)(p))

struct RayCull {
    V3 inv, nod;  // 1/d and -(o/d)
    // Byte offsets inside a BvhNode4 of the planes the ray ENTERS through, per axis (lo_* where 1/d is
    // positive, hi_* otherwise); the exit planes are at offset ^ {48, 80, 112}.
    uint32_t near_x, near_y, near_z;
    float pad_base, pad_k;
};

__device__ __forceinline__ RayCull make_cull(V3 o, V3 d, const float* center, float radius, float eps_frac) {
    RayCull rc;
    const V3 oc = o - mk(center);
    const float S = __builtin_amdgcn_sqrtf(dot(oc, oc)) + radius;
    const float omax =
        __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z));
    rc.pad_base = (64.0f / 16777216.0f) * (S + omax);
    rc.pad_k = rc.pad_base * (__builtin_amdgcn_sqrtf(dot(d, d)) * eps_frac);
    rc.inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    rc.nod = mk(-(o.x * rc.inv.x), -(o.y * rc.inv.y), -(o.z * rc.inv.z));
    rc.near_x = (__float_as_uint(rc.inv.x) >> 31) ? 48u : 0u;
    rc.near_y = (__float_as_uint(rc.inv.y) >> 31) ? 64u : 16u;
    rc.near_z = (__float_as_uint(rc.inv.z) >> 31) ? 80u : 32u;
    return rc;
}

constexpr uint32_t kMissKey = 0xFFFFFFFFu;

// One child box of a 4-wide node, given its entry planes (nx, ny, nz) and exit planes (fx, fy, fz) for
// this ray's direction signs. Growing the box by the pad moves entry planes against the ray and exit planes
// with it; in the ray-parameter domain that is a constant per axis, folded into the FMA's addend by the
// caller (nodn = nod - pad|1/d|, nodf = nod + pad|1/d|), so one FMA per plane gives the padded plane's
// parameter. Returns the sort key of the child: kMissKey when the ray misses it, else its entry distance
// (two low mantissa bits replaced by the child slot).
// The three conditions of a hit -- tn <= tf, tf >= eps, tn <= best_t -- are ONE compare: with best_t >= eps (every
// search bound is: triangle.rs:398's 1e6, a sphere's distance relaxed upwards, an accepted t > eps) they are
// equivalent to max(tn, eps) <= min(tf, best_t). fmax / fmin ignore a NaN operand, so a NaN plane parameter (a
// zero direction component) can only make the test MORE permissive, which culling tolerates; an UNUSED slot has the
// empty box (lo = +inf, hi = -inf: its entry parameter is +inf or NaN on every axis, its exit parameter -inf or
// NaN) and misses unless all six are NaN, i.e. unless the ray itself is degenerate -- those rays never get here
// (ray_is_traversable below).
template <bool SLOT_BITS = true>
__device__ __forceinline__ uint32_t child_key(float nx, float ny, float nz, float fx, float fy, float fz, V3 inv, V3 nodn,
                                              V3 nodf, float eps, float best_t, uint32_t slot) {
    const float tnx = __builtin_fmaf(nx, inv.x, nodn.x);
    const float tny = __builtin_fmaf(ny, inv.y, nodn.y);
    const float tnz = __builtin_fmaf(nz, inv.z, nodn.z);
    const float tfx = __builtin_fmaf(fx, inv.x, nodf.x);
    const float tfy = __builtin_fmaf(fy, inv.y, nodf.y);
    const float tfz = __builtin_fmaf(fz, inv.z, nodf.z);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz), eps);
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(tfx, tfy), tfz), best_t);
    // (tn >= eps > 0: the integer order of its bits is its order; a caller that sorts the links along with the keys has no
    // use for the slot number in the low bits)
    return tn <= tf ? (SLOT_BITS ? ((__float_as_uint(tn) & ~3u) | slot) : __float_as_uint(tn)) : kMissKey;
}
// A ray the slab arithmetic above can be trusted with: finite origin, finite non-zero direction. Anything else gets
// no mesh hit at all, which is what the reference's ordered compares give it (triangle.rs:190-241: every NaN compare
// is false) -- callers start such a ray's search at a bound below eps, so that every node test fails.
__device__ __forceinline__ bool ray_is_traversable(V3 o, V3 d) {
    const float dd = dot(d, d), oo = dot(o, o);
    return dd > 0.0f && dd < __builtin_inff() && oo < __builtin_inff();  // (false for NaN)
}

__device__ __forceinline__ void cswap(uint32_t& a, uint32_t& b) {
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}

// Fetch one 128-B node (8 x dwordx4 by this lane) and test its four children; k[] comes back sorted by
// entry distance (misses last) -- or, with SORTED = false, in slot order --, links = the four child links.
template <bool SORTED = true, bool SLOT_BITS = true>
__device__ __forceinline__ void node4_visit(const BvhNode4* node, const RayCull& rc, float eps, float best_t,
                                            uint32_t k[4], f32x4& links) {
    const char* nb = reinterpret_cast<const char*>(node);
    f32x4 nx, ny, nz, fx, fy, fz, me;
    nx = *RBRT_AS1(f32x4, nb + rc.near_x), ny = *RBRT_AS1(f32x4, nb + rc.near_y), nz = *RBRT_AS1(f32x4, nb + rc.near_z);
    fx = *RBRT_AS1(f32x4, nb + (rc.near_x ^ 48u)), fy = *RBRT_AS1(f32x4, nb + (rc.near_y ^ 80u));
    fz = *RBRT_AS1(f32x4, nb + (rc.near_z ^ 112u));
    me = *RBRT_AS1(f32x4, nb + 112);
    links = *RBRT_AS1(f32x4, nb + 96);
    // one pad for the node: the largest of its children's error terms (siblings have similar triangles)
    // (the error terms are products of lengths, >= +0: the order of their bits as integers is their order, and an integer
    // max3 needs no canonicalising v_max x, x, x in front of it, which fmaxf of a loaded value gets. A NaN -- a mesh with
    // non-finite vertices -- comes out as the maximum and makes every test of the node pass.)
    const uint32_t me_max = max(max(__float_as_uint(me.x), __float_as_uint(me.y)), max(__float_as_uint(me.z), __float_as_uint(me.w)));
    const float pad = __builtin_fmaf(rc.pad_k, __uint_as_float(me_max), rc.pad_base);
    // the pad in ray-parameter units, per axis (|1/d| may be inf: the plane parameters are then +-inf or NaN,
    // which only makes the test more permissive)
    const V3 pt = mk(pad * __builtin_fabsf(rc.inv.x), pad * __builtin_fabsf(rc.inv.y), pad * __builtin_fabsf(rc.inv.z));
    const V3 nodn = rc.nod - pt, nodf = rc.nod + pt;
    k[0] = child_key<SLOT_BITS>(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, rc.inv, nodn, nodf, eps, best_t, 0u);
    k[1] = child_key<SLOT_BITS>(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, rc.inv, nodn, nodf, eps, best_t, 1u);
    k[2] = child_key<SLOT_BITS>(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, rc.inv, nodn, nodf, eps, best_t, 2u);
    k[3] = child_key<SLOT_BITS>(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, rc.inv, nodn, nodf, eps, best_t, 3u);
    if (SORTED) {
        cswap(k[0], k[1]);
        cswap(k[2], k[3]);
        cswap(k[0], k[2]);
        cswap(k[1], k[3]);
        cswap(k[1], k[2]);
    }
}
// The same visit with the child LINKS sorted along with the keys (round 4). Sorting the keys alone costs two
// instructions per compare-exchange (v_min_u32 + v_max_u32), but every link then has to be picked out of four registers
// by the slot number in its key's low bits: and + 3 x (v_cmp, v_cndmask) = 7 VALU instructions and, with the compares
// writing VCC right ahead of the selects that read it, three s_nop each -- 28 + 12 per visit for the four links. A
// compare-exchange that carries the link along is one compare and four selects: 25 for the five exchanges, against
// 10 + 28, and five VCC hazards instead of twelve. Which child is walked first cannot change what is found (the tree only
// culls; the result cell keeps the lexicographic (t, index) minimum), so equal keys may come out in either order.
#ifndef RBRT_PAIR_SORT
#define RBRT_PAIR_SORT 1
#endif
__device__ __forceinline__ void cswap_pair(uint32_t& ka, uint32_t& kb, int32_t& la, int32_t& lb) {
    const bool sw = kb < ka;
    const uint32_t k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const int32_t l0 = sw ? lb : la, l1 = sw ? la : lb;
    ka = k0, kb = k1, la = l0, lb = l1;
}
__device__ __forceinline__ void node4_visit_sorted(const BvhNode4* node, const RayCull& rc, float eps, float best_t,
                                                   uint32_t k[4], int32_t l[4]) {
    f32x4 links;
    node4_visit<false, false>(node, rc, eps, best_t, k, links);
    l[0] = __float_as_int(links.x), l[1] = __float_as_int(links.y), l[2] = __float_as_int(links.z), l[3] = __float_as_int(links.w);
    cswap_pair(k[0], k[1], l[0], l[1]);
    cswap_pair(k[2], k[3], l[2], l[3]);
    cswap_pair(k[0], k[2], l[0], l[2]);
    cswap_pair(k[1], k[3], l[1], l[3]);
    cswap_pair(k[1], k[2], l[1], l[2]);
}
__device__ __forceinline__ int32_t link_of(const f32x4& links, uint32_t key) {
    const uint32_t s = key & 3u;
    const float v = s == 0u ? links.x : (s == 1u ? links.y : (s == 2u ? links.z : links.w));
    return __float_as_int(v);
}

// All triangles of one leaf (<= kLeafMax), triangle.rs:189-255 + the arg-min rule of triangle.rs:400 (the simple
// one-thread-per-ray path; the megakernel deals the triangles of pending leaves out to all lanes instead).
template <bool STATS>
__device__ __forceinline__ void leaf_test(const BvhTri* tris, int32_t leaf_ref, V3 o, V3 d, float eps, float eps_frac,
                                          float& best_t, uint32_t& best_idx, LocalCounters& lc) {
    const uint32_t leaf = uint32_t(~leaf_ref);
    const uint32_t first = leaf >> kLeafBits, last = leaf & uint32_t(kLeafMax - 1);  // last = count - 1
    for (uint32_t i = 0; i <= last; ++i) {
        const auto* tp = RBRT_AS1(f32x4, tris + first + i);
        const f32x4 a = tp[0], b = tp[1];
        const f32x2 c = *RBRT_AS1(f32x2, tp + 2);  // the record's last 8 bytes are padding: not fetched
        if (STATS) ++lc.tris;
        float t;
        const bool hit = tri_test(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), o, d, eps, eps_frac, t);
        const uint32_t idx = __float_as_uint(c.y);
        // strict < keeps the first (lowest) index among equal t
        if (hit && (t < best_t || (t == best_t && idx < best_idx))) {
            best_t = t;
            best_idx = idx;
        }
    }
}

// Closest accepted triangle of one mesh = triangle.rs:134-262 + 392-410, found through the BVH
// (simple one-thread-per-ray form with the whole stack in LDS; the megakernel has its own loop).
// Result contract (what the brute-force scan returns): the smallest accepted t below the scan's
// initial 1e6; among equal t the lowest reference index. best_t stays 1e6 when nothing is hit.
template <bool STATS>
__device__ __forceinline__ void mesh_closest(const DevMesh& M, V3 o, V3 d, float eps, float eps_frac,
                                             uint32_t* __restrict__ stack, uint32_t stride, float& best_t_out,
                                             uint32_t& best_idx_out, LocalCounters& lc) {
    const RayCull rc = make_cull(o, d, M.center, M.radius, eps_frac);
    float best_t = 1000000.0f;  // triangle.rs:398
    uint32_t best_idx = 0;
    int sp = 0;
    int32_t cur = ray_is_traversable(o, d) ? 0 : kNoChild;  // root
    while (cur != kNoChild) {
        bool pop = true;
        if (cur >= 0) {
            uint32_t k[4];
            f32x4 links;
            node4_visit(M.nodes + cur, rc, eps, best_t, k, links);
            if (STATS) ++lc.nodes;
            if (k[0] != kMissKey) {
                if (k[3] != kMissKey) stack[(sp++) * stride] = uint32_t(link_of(links, k[3]));
                if (k[2] != kMissKey) stack[(sp++) * stride] = uint32_t(link_of(links, k[2]));
                if (k[1] != kMissKey) stack[(sp++) * stride] = uint32_t(link_of(links, k[1]));
                cur = link_of(links, k[0]);
                pop = false;
            }
        } else {
            leaf_test<STATS>(M.tris, cur, o, d, eps, eps_frac, best_t, best_idx, lc);
        }
        if (pop) {
            if (sp == 0) break;
            cur = int32_t(stack[(--sp) * stride]);
        }
    }
    best_t_out = best_t;
    best_idx_out = best_idx;
}

struct HitRec {
    float dist;   // dist_from_ray_orig of the winner (lib.rs:35)
    float t;      // its ray parameter
    int32_t obj;  // -1 miss, [0, n_elem) an element of Scene::elements (sphere or BasicTriangle), n_elem + m mesh
    uint32_t tri;
};

// scene.rs:19-43: spheres in order, then meshes in order, strictly smaller distance wins.
template <bool STATS>
__device__ __forceinline__ void scene_hit(const TraceParams& P, V3 o, V3 d, uint32_t* stack, uint32_t stride,
                                          HitRec& h, LocalCounters& lc) {
    float closest = 3.40282347e+38f;  // f32::MAX
    h.obj = -1;
    h.t = 0.0f;
    h.tri = 0;
    if (STATS) ++lc.rays;
    const uint32_t n_elem = P.n_spheres + P.n_elem_tris;
    for (uint32_t e = 0; e < n_elem; ++e) {  // Scene::elements in their order (scene.rs:23-31)
        const uint32_t desc = P.elems != nullptr ? P.elems[e] : e;
        float t, dist;
        bool hit;
        if (desc >> 31) {
            hit = basic_triangle_hit(P.elem_tris[desc & 0x7FFFFFFFu].v0, o, d, P.min_dist, P.max_dist, t, dist);
        } else {
            const DevSphere sp = P.spheres[desc];
            hit = sphere_hit(mk(sp.center), sp.radius, o, d, P.min_dist, P.max_dist, t, dist, P.counters);
        }
        if (hit && dist < closest) {
            closest = dist;
            h.t = t;
            h.obj = int32_t(e);
        }
    }
    for (uint32_t m = 0; m < P.n_meshes; ++m) {
        const DevMesh& M = P.meshes[m];
        if (!bbox_gate(M.bbox_lo, M.bbox_hi, o, d)) continue;  // mesh.rs:233-235
        if (STATS) ++lc.gate;
        float t;
        uint32_t idx;
        mesh_closest<STATS>(M, o, d, P.min_dist, P.eps_frac, stack, stride, t, idx, lc);
        if (t > P.min_dist && t < 100000.0f) {  // triangle.rs:405
            V3 p = o + t * d;                    // mesh.rs:247-249
            float dist = length(o - p);
            if (dist > P.min_dist && dist < P.max_dist) {
                if (STATS) ++lc.mesh_hits;
                if (dist < closest) {
                    closest = dist;
                    h.t = t;
                    h.obj = int32_t(n_elem + m);
                    h.tri = idx;
                }
            }
        }
    }
    h.dist = closest;
}

// ---------------------------------------------------------------------------------------------
// Materials
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 random_point_in_unit_sphere(Rng& rng) {  // materials.rs:14-30
    // The reference loops `while point.length() > 1.0`. With a correctly rounded sqrt,
    // sqrt(s) > 1.0f holds exactly when s > 0x1.000002p0f (the float after 1.0): sqrt(1 + 2^-23)
    // still rounds to 1.0, sqrt(1 + 2^-22) rounds to 1 + 2^-23. So the loop can compare the squared
    // length and skip an IEEE square root per iteration with bit-identical accept/reject decisions
    // (checked exhaustively around 1.0 in tests/test_oracle_kats.py).
    V3 p;
    float s;
    do {
        float x = rng.next_f32();
        float y = rng.next_f32();
        float z = rng.next_f32();
        p = 2.0f * mk(x, y, z) - mk(1.0f, 1.0f, 1.0f);
        s = (p.x * p.x + p.y * p.y) + p.z * p.z;
    } while (s > 1.00000011920928955078125f);
    return p;
}

__device__ __forceinline__ V3 reflect(V3 dir, V3 n) {  // materials.rs:32-37
    V3 du = normalize(dir);
    V3 nu = normalize(n);
    V3 r = du - 2.0f * nu * dot(du, nu);
    return normalize(r);
}

__device__ __forceinline__ float schlick(float cosine, float ref_index) {  // dielectric.rs:63-66
    float q = (1.0f - ref_index) / (1.0f + ref_index);
    float r0 = q * q;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x4 = x2 * x2;
    return r0 + (1.0f - r0) * (x * x4);  // powi(5) = x * (x^2)^2
}

__device__ __forceinline__ bool refract(V3 dir, V3 n, float ni_over_nt, V3& out) {  // dielectric.rs:68-85
    V3 v = normalize(dir);
    V3 nu = normalize(n);
    float c = dot(v, nu);
    float discr = 1.0f - (ni_over_nt * ni_over_nt) * (1.0f - c * c);
    if (discr > 0.0f) {
        out = ni_over_nt * (v - nu * c) - __builtin_sqrtf(discr) * nu;
        return true;
    }
    return false;
}

// RayScattering::scatter for the three materials. Returns the reference's bool; `records`
// says whether the attenuation is anything other than the exact identity (1,1,1).
__device__ __forceinline__ bool scatter(const DevMaterial& m, V3 in_d, V3 p, V3 n, Rng& rng, V3& out_d) {
    if (m.kind == RBRT_MAT_LAMBERTIAN) {  // lambertian.rs:11-24
        V3 target = (p + normalize(n)) + random_point_in_unit_sphere(rng);
        out_d = normalize(target - p);
        return true;
    } else if (m.kind == RBRT_MAT_METAL) {  // metal.rs:12-25
        V3 target = reflect(in_d, n);
        out_d = normalize(target + m.param * random_point_in_unit_sphere(rng));
        return dot(out_d, n) > 0.0f;
    } else {  // dielectric.rs:11-59
        V3 reflected = reflect(in_d, n);
        V3 outward;
        float ni_over_nt, cosine;
        float a = dot(normalize(in_d), normalize(n));
        if (a > 0.0f) {
            outward = -1.0f * n;
            ni_over_nt = m.param;
            cosine = m.param * a;
        } else {
            outward = n;
            ni_over_nt = 1.0f / m.param;
            cosine = -a;
        }
        V3 refracted = mk(0.0f, 0.0f, 0.0f);
        float reflect_prob = refract(in_d, outward, ni_over_nt, refracted) ? schlick(cosine, m.param) : 1.0f;
        out_d = (rng.next_f32() < reflect_prob) ? reflected : refracted;
        return true;
    }
}

// cam.rs:64-82: direction of the camera ray of sample (pixel, stream state); the column jitter is drawn first.
__device__ __forceinline__ V3 camera_ray_direction(V3 pos, V3 center, V3 right, V3 up, float mm_hor, float mm_vert, uint32_t img_w,
                                                   uint32_t img_h, uint32_t row, uint32_t col, Rng& rng) {
    const float col_off = float(col) - float(img_w / 2);
    const float row_off = float(row) - float(img_h / 2);
    const float u0 = rng.next_f32();
    const float col_mm = ((col_off + u0) - 0.5f) * mm_hor;
    const float u1 = rng.next_f32();
    const float row_mm = ((row_off + u1) - 0.5f) * mm_vert;
    const V3 target = (center + (0.001f * col_mm) * right) - (0.001f * row_mm) * up;
    return normalize(target - pos);
}

// lib.rs:68-71: what a ray that hits nothing sees; the direction as it is (not re-normalised).
__device__ __forceinline__ V3 background(float dy, const float* bg) {
    const float t = 0.5f * (dy + 1.0f);
    return t * mk(1.0f, 1.0f, 1.0f) + (1.0f - t) * mk(bg);
}

__host__ __device__ inline uint32_t megakernel_lds_dwords(uint32_t pool, uint32_t stack_entries, uint32_t n_spheres, uint32_t n_meshes,
                                                          uint32_t n_elem_tris);
#include "megakernel.inl"

// lib.rs:116-122: (sqrt(c) * 256) as u8 — Rust's float->int cast saturates and maps NaN to 0.
__device__ __forceinline__ uint8_t quantise(float c) {
    float v = __builtin_sqrtf(c) * 256.0f;
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return uint8_t(v);
}

// The end of a pixel's batch: the running sum goes back to `acc`, or -- on the last batch -- the mean (lib.rs:101) and
// the optional quantisation go out. `j`: the pixel's place in the rank's packed tiles.
__device__ __forceinline__ void resolve_store(const ResolveParams& R, size_t j, uint32_t row, uint32_t col, bool valid, float ax, float ay,
                                              float az) {
    const bool packed = R.tile_world > 1;
    if (valid) {
        if (!R.last_batch) {
            R.acc[j * 3 + 0] = ax;
            R.acc[j * 3 + 1] = ay;
            R.acc[j * 3 + 2] = az;
            return;
        }
        ax = ax * R.inv_spp;
        ay = ay * R.inv_spp;
        az = az * R.inv_spp;
    } else if (!R.last_batch || !packed) {
        return;  // (a pixel slot beyond a ragged image edge: packed outputs carry it as zeros)
    }
    const size_t o = packed ? j * 3u : (size_t(row) * R.width + col) * 3u;
    if (R.out_radiance) {
        R.out_radiance[o + 0] = ax;
        R.out_radiance[o + 1] = ay;
        R.out_radiance[o + 2] = az;
    }
    if (R.out_rgb8) {
        R.out_rgb8[o + 0] = quantise(ax);
        R.out_rgb8[o + 1] = quantise(ay);
        R.out_rgb8[o + 2] = quantise(az);
    }
}

// Sequential per-pixel sum over the samples of this batch (lib.rs:95-100 adds them in sample
// order; keeping that order keeps the mean bit-identical), then on the last batch the multiply by
// 1/spp (lib.rs:101) and the optional quantisation.
__global__ __launch_bounds__(kSmallBlock) void resolve_kernel(const ResolveParams R) {
    // (with a tile pass the launch rendered the tiles of the work list only; the grid covers all of the rank's)
    const size_t npix = size_t(R.tile_lists ? R.tile_lists[0] : R.n_local_tiles) * 64u;
    const size_t i = size_t(blockIdx.x) * kSmallBlock + threadIdx.x;
    // the trace launch this resolves has ended: its work counters are reset for the lane's next launch (which
    // waits for this kernel), saving a memset launch that would have to queue behind the resident megakernels
    // Helper launches (api.cpp "Elastic launches") may have joined the launch: a helper wave that took work registered in
    // helper_words[0] before it drew from the counters, and leaves the count only when its samples are written. The launch
    // itself has ended (stream order), so nobody can TAKE work any more; what is waited for here is the waves that hold
    // some. Almost always the count is zero already. Bounded: a count that never returns raises the error flag instead of
    // hanging the GPU (rbrt_hip_scene_check reports it).
    if (R.helper_words) {
        // (the common case is a count of zero, read with a plain atomic load: an acquire here would invalidate the CU's L1
        // under the resident trace waves, once per workgroup of this kernel -- 12,288 times per 1024x768 frame: +3-5 % per
        // frame, measured. A count of zero means every helper wave that held work has written its samples back and left;
        // this kernel's caches were invalidated when it started and none of its waves reads a sample before this point.)
        if (threadIdx.x == 0 && __hip_atomic_load(R.helper_words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            uint32_t spins = 0;
            while (__hip_atomic_load(R.helper_words, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1u << 23)) {
                    atomicAdd(R.error_flag, 1ull);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        // (first the launch's number -- from here on a helper wave of it draws nothing more --, then the counters)
        if (R.helper_words && threadIdx.x == 0) {
            __hip_atomic_store(R.helper_words + 1, R.helper_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < kWorkShards * kWorkCounterStride; w += kSmallBlock) R.work_counter[w] = 0ull;
    }
    if (i >= npix) return;
    const uint32_t p = uint32_t(i & 63u);
    const uint32_t tile_local = R.tile_lists ? R.tile_lists[kTileListHeader + (i >> 6)] : uint32_t(i >> 6);
    const size_t j = size_t(tile_local) * 64u + p;  // the pixel's place in the rank's packed tiles (acc, packed outputs)
    const uint32_t tile = tile_local * R.tile_world + R.tile_rank;
    uint32_t ty, tx;
    tile_xy(tile, R.tiles_x, ty, tx);
    const uint32_t row = ty * RBRT_TILE + (p >> 3), col = tx * RBRT_TILE + (p & 7u);
    const bool valid = row < R.height && col < R.width;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    if (valid) {
        if (!R.first_batch) {
            ax = R.acc[j * 3 + 0];
            ay = R.acc[j * 3 + 1];
            az = R.acc[j * 3 + 2];
        }
        for (uint32_t s = 0; s < R.batch; ++s) {
            const float* sp = R.sample_buf + (size_t(s) * npix + i) * 3u;
            ax = ax + sp[0];
            ay = ay + sp[1];
            az = az + sp[2];
        }
    }
    resolve_store(R, j, row, col, valid, ax, ay, az);
}

// The pixels of the tiles that see only the background (TraceParams::tile_lists, second list): every sample of such a
// pixel is one camera ray that hits nothing (cam.rs:64-82, lib.rs:68-71), so its batch is generated, summed in sample
// order and stored right here -- the same streams, the same arithmetic, the same order as the trace kernel followed by
// resolve_kernel would produce -- and the trace kernel never sees these tiles. One thread per pixel.
__global__ __launch_bounds__(kSmallBlock) void sky_resolve_kernel(const TraceParams P, const ResolveParams R) {
    const size_t i = size_t(blockIdx.x) * kSmallBlock + threadIdx.x;
    if (i >= size_t(R.tile_lists[1]) * 64u) return;
    const uint32_t p = uint32_t(i & 63u);
    const uint32_t tile_local = R.tile_lists[kTileListHeader + R.n_local_tiles + (i >> 6)];
    const size_t j = size_t(tile_local) * 64u + p;
    const uint32_t tile = tile_local * R.tile_world + R.tile_rank;
    uint32_t ty, tx;
    tile_xy(tile, R.tiles_x, ty, tx);
    const uint32_t row = ty * RBRT_TILE + (p >> 3), col = tx * RBRT_TILE + (p & 7u);
    const bool valid = row < R.height && col < R.width;
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    if (valid) {
        if (!R.first_batch) {
            ax = R.acc[j * 3 + 0];
            ay = R.acc[j * 3 + 1];
            az = R.acc[j * 3 + 2];
        }
        const V3 pos = mk(P.cam.position), center = mk(P.cam.img_center_point), right = mk(P.cam.right), up = mk(P.cam.up);
        for (uint32_t s = 0; s < R.batch; ++s) {
            Rng rng;
            rng.init(P.seed_key, row * R.width + col, P.sample_base + s);
            const V3 d = camera_ray_direction(pos, center, right, up, P.cam.mm_per_pix_hor, P.cam.mm_per_pix_vert, R.width, R.height, row, col, rng);
            const V3 c = background(d.y, P.bg);
            ax = ax + c.x;
            ay = ay + c.y;
            az = az + c.z;
        }
        if (R.counters) {  // a counting launch: one ray, one sample, zero bounces each
            atomicAdd(&R.counters->rays, (unsigned long long)R.batch);
            atomicAdd(&R.counters->samples, (unsigned long long)R.batch);
            atomicAdd(&R.counters->diag[32], (unsigned long long)R.batch);
        }
    }
    resolve_store(R, j, row, col, valid, ax, ay, az);
}

// ---------------------------------------------------------------------------------------------
// The tile pass (DESIGN.md "The tile pass"). Every camera ray of an 8x8 tile starts at cam.position and passes through
// a small rectangle of the image plane (cam.rs:64-82), so which spheres and meshes the tile's rays can reach at all is
// known before a single ray is made. This kernel writes one word per tile (TraceParams::tile_cull); a tile whose rays
// can reach NOTHING sees the background only (lib.rs:68-71): tile_lists_kernel takes it off the trace kernel's work and
// sky_resolve_kernel finishes its pixels. A test declared out of reach is one the reference would have run and FAILED,
// so the image does not change; the proof obligation is that the rule never rules out a test that could succeed.
//
// Directions. The tile's exact ray directions are the unit vectors from the position to a (widened) rectangle of the
// image plane: a convex spherical quadrilateral, inside the cone of half-angle rho (largest angle to a corner) around
// the direction c of its middle. A sphere of radius R at distance l > R fills the cone of half-angle asin(R / l) around
// q = (centre - position) / l. So with the OUTER radius R_out (inflated, below):
//   forward rays miss it     when angle(c, q)  > asin(R_out / l) + rho,
//   backward lines miss it   when angle(-c, q) > asin(R_out / l) + rho,
//   backward lines cut DEEP  when all four corners k have angle(-corner_k, q) < asin(R_in / l) (R_in: deflated; a cap of
//                            less than a quarter turn is convex, so the corners speak for the tile; the position clearly outside).
// sphere.rs:20-66 needs the forward condition and one of the backward ones: it looks at the LINE first (discriminant),
// and with a discriminant of exactly zero it accepts a negative parameter (sphere.rs:41-49: the `sol > 0.0 &&` guard),
// so a line that grazes the sphere BEHIND the camera could be a hit; one that misses it there cannot, and one that cuts
// deep has a discriminant well above zero, takes the guarded branch and is rejected with both roots negative (b > 0 and
// 4 a c >= 8e-3 r^2 keep the larger root negative in float as well). This is what lets the sky tiles drop the ground
// sphere, which every line through the camera cuts somewhere. The box test (aabbox.rs:28-58) rejects `t_max < 0`
// first, so for a mesh the forward condition on the box's bounding sphere is enough.
//
// A second, independent rule catches long thin cases the cones cannot: with a = img_center - position, f_x(p) =
// [a + 0.001 x right, up, p - position] (triple product) vanishes on the plane of the camera lines of column
// coordinate x, and a point of the line of coordinate cm at parameter s has f_x = s 0.001 (cm - x) [a, up, right]:
// over the tile's interval [lo, hi] every point of every one of its lines has f_lo f_hi <= 0, whatever the signs. An
// object entirely inside {f_lo > 0, f_hi > 0} or {f_lo < 0, f_hi < 0} meets none of the lines at all. Same with rows.
//
// Margins. The kernel's rays are float: the target point carries an absolute error of a few ulp of |img_center| and
// |position| (target - position cancels), an angle of err / |target - position| against the exact line (taken as 16
// unit roundoffs of the magnitudes involved, twice over). The sphere test with u = 2^-24, a = d.d = 1 +- 4u: l = o - c
// carries u |l| per component, b = (2d).l at most 8u |l|, c = l.l - r r at most 6u (|l|^2 + r^2), and the discriminant
// b b - (4a) c at most 36u |l|^2 + 48u (|l|^2 + r^2) + 4u max(b^2, 4ac) <= 90u (|l|^2 + r^2) = 5.4e-6 (|l|^2 + r^2)
// against its exact value 4 (r^2 - m^2), m the distance of the line from the centre: a line is taken for a hit at most
// 0.7e-6 (|l|^2 + r^2) / r outside the sphere, and a line more than that inside has a discriminant above zero. Radii are
// inflated (deflated) by 1e-5 (|l|^2 + r^2) / r, 15x that, plus 1e-5 r + 1e-6 |l|; angles are widened by the direction
// error, the pixel interval by 1e-3 pixel plus the rounding of (col_off + u) - 0.5. The mesh boxes' slack is derived where
// it is used (16 u of the farthest corner for the box test itself; the traversal's own pad for the boxes of the tree). (A first version used 0.1 % of the
// radius: a full unit for the ground sphere, which made the band of tiles whose backward lines "might graze" it six tile
// rows high instead of two.) A degenerate camera makes the quantities NaN and
// every comparison false: nothing is culled. Non-finite or non-positive radii and non-finite boxes (a mesh without
// triangles has lo = +inf) are never culled: the reference's NaN panic (sphere.rs:33), reported through
// nan_discriminants, must still be reached. BasicTriangle elements are never culled (the YAML cannot describe them).
// ---------------------------------------------------------------------------------------------
struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 d3(const float* p) { return D3{double(p[0]), double(p[1]), double(p[2])}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return D3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 dcross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dlen(D3 a) { return sqrt(ddot(a, a)); }
__device__ __forceinline__ D3 unit_or_nan(D3 v) { return (1.0 / dlen(v)) * v; }  // (a zero vector gives NaN: nothing is culled with it)
__device__ __forceinline__ double angle_between(D3 u, D3 v) {  // unit vectors; NaN stays NaN
    const double c = ddot(u, v);
    return acos(c > 1.0 ? 1.0 : c < -1.0 ? -1.0 : c);
}

// Two planes through the camera position (unit normals) bounding the camera lines of a pixel interval.
struct CullSlab {
    D3 n_lo, n_hi;
    // every point within `reach` of q (relative to the position) strictly on one and the same side of both planes?
    __device__ __forceinline__ bool outside(D3 q, double reach) const {
        const double g0 = ddot(n_lo, q), g1 = ddot(n_hi, q);
        return (g0 > reach && g1 > reach) || (g0 < -reach && g1 < -reach);  // (false for NaN)
    }
};
// The tile's ray directions: inside the cone of half-angle rho around mid (unit). The angle comparisons of the header are
// made on cosines (one square root per ball instead of an arc cosine and an arc sine): for t in [0, pi] and 0 <= a + rho
// <= pi, t > a + rho is cos t < cos a cos rho - sin a sin rho, with sin a = R / l.
struct CullCone {
    D3 mid;
    double cos_rho, sin_rho;  // rho < pi / 2 (else NaN: nothing is culled)
    // forward rays (sign +1) or backward extensions (sign -1) all miss the ball of radius R at q = l * qhat?
    __device__ __forceinline__ bool misses(D3 qhat, double l, double R, double sign) const {
        if (!(R < l)) return false;
        const double sa = R / l, ca = sqrt(1.0 - sa * sa);
        return sign * ddot(mid, qhat) < ca * cos_rho - sa * sin_rho;  // (false for NaN)
    }
    // (a cap of less than a quarter turn is convex on the sphere: the four corners inside it put the whole tile inside)
    D3 corner[4];
    double cos_w, sin_w;  // w: the float ray's direction error
    __device__ __forceinline__ bool backward_deep(D3 qhat, double l, double r_in) const {
        if (!(r_in > 0.0 && r_in < l)) return false;
        const double sa = r_in / l, ca = sqrt(1.0 - sa * sa);
        if (!(sa > sin_w)) return false;                 // the cap asin(r_in / l) - w must be a cap
        const double cos_cap = ca * cos_w + sa * sin_w;  // cos(asin(r_in / l) - w)
        bool in = true;
        for (uint32_t k = 0; k < 4u; ++k) in = in && -ddot(corner[k], qhat) > cos_cap;
        return in;
    }
};

// FOUR lanes per tile (round 4). The kernel's time was the longest walk over the top of a mesh's tree -- a tile at the
// mesh's silhouette tests the four child boxes of some twenty nodes one after the other, in double precision, a chain of
// ~40,000 dependent operations: 0.12 ms for 12,288 tiles, in front of every frame with a new camera. The four lanes of a
// tile compute the same prologue and the same sphere bits (cheap, redundant) and share the walk: every lane pops the same
// node and tests ONE of its four children, a wave vote brings the quad's four answers to each of its lanes, and every
// lane pushes the same entries on its own copy of the stack (no LDS, no hand-over).
constexpr int kCullBlock = 64;       // one wave per workgroup: the launch spreads over the CUs
constexpr uint32_t kCullLanes = 4;   // lanes per tile
// (at most 128 VGPRs, like a trace wave: a wave of this kernel then fits the slot ONE exiting trace wave leaves on its SIMD;
// at 140 it needed two of them to have left)
__global__ __launch_bounds__(kCullBlock, 4) void primary_cull_kernel(const TraceParams P) {
    // (a rank of a multi-GPU render makes the words of ITS tiles only -- tile = local * world + rank, the only words its
    // lists and its launch read: an eighth of the pass for an eighth of the frame)
    const uint32_t cull_world = P.tile_world > 1u ? P.tile_world : 1u;
    const uint32_t tile_local = (blockIdx.x * kCullBlock + threadIdx.x) / kCullLanes;
    const uint32_t tile = tile_local * cull_world + (cull_world > 1u ? P.tile_rank : 0u);
    const uint32_t quad_lane = threadIdx.x % kCullLanes, quad_shift = threadIdx.x & ~(kCullLanes - 1u);
    if (tile >= P.n_tiles) return;  // (whole quads: the votes below count active lanes only)
    uint32_t ty, tx;
    tile_xy(tile, P.tiles_x, ty, tx);
    const rbrt_camera_t& c = P.cam;
    const uint32_t W = c.img_width_pix, H = c.img_height_pix;
    const D3 pos = d3(c.position), right = d3(c.right), up = d3(c.up), ctr = d3(c.img_center_point);
    const D3 a = ctr - pos;
    // the tile's pixel interval in the units of cam.rs:70-75, widened
    const double pad = 1e-3 + double(W > H ? W : H) * (1.0 / 4194304.0);
    const uint32_t c0 = tx * RBRT_TILE, r0 = ty * RBRT_TILE;
    const uint32_t c1 = (c0 + RBRT_TILE < W ? c0 + RBRT_TILE : W) - 1u, r1 = (r0 + RBRT_TILE < H ? r0 + RBRT_TILE : H) - 1u;
    const double mmh = double(c.mm_per_pix_hor), mmv = double(c.mm_per_pix_vert);
    const double cm0 = (double(c0) - double(W / 2u) - 0.5 - pad) * mmh, cm1 = (double(c1) - double(W / 2u) + 0.5 + pad) * mmh;
    const double rm0 = (double(r0) - double(H / 2u) - 0.5 - pad) * mmv, rm1 = (double(r1) - double(H / 2u) + 0.5 + pad) * mmv;
    CullSlab cols, rows;
    cols.n_lo = unit_or_nan(dcross(a + (0.001 * cm0) * right, up));
    cols.n_hi = unit_or_nan(dcross(a + (0.001 * cm1) * right, up));
    rows.n_lo = unit_or_nan(dcross(a - (0.001 * rm0) * up, right));
    rows.n_hi = unit_or_nan(dcross(a - (0.001 * rm1) * up, right));
    // the float ray against the exact line: absolute error of (target - position), as an angle
    const double cm_abs = fabs(cm0) > fabs(cm1) ? fabs(cm0) : fabs(cm1), rm_abs = fabs(rm0) > fabs(rm1) ? fabs(rm0) : fabs(rm1);
    const double err = 8.0 * (1.0 / 8388608.0) * (dlen(ctr) + dlen(pos) + 0.001 * cm_abs * dlen(right) + 0.001 * rm_abs * dlen(up));
    const double plane_dist = fabs(ddot(a, unit_or_nan(dcross(right, up))));  // no target is closer to the position
    const double angle = 2.0 * err / plane_dist + 1e-6;                          // (NaN or inf for a degenerate camera)
    CullCone cone;
    cone.mid = unit_or_nan(a + (0.0005 * (cm0 + cm1)) * right - (0.0005 * (rm0 + rm1)) * up);
    double rho = 0.0;
    for (uint32_t k = 0; k < 4u; ++k) {
        const D3 u = unit_or_nan(a + (0.001 * (k & 1u ? cm1 : cm0)) * right - (0.001 * (k & 2u ? rm1 : rm0)) * up);
        cone.corner[k] = u;
        const double t = angle_between(cone.mid, u);
        rho = t > rho || !(t == t) ? t : rho;  // (a NaN sticks)
    }
    rho += angle;
    if (!(rho < 1.5)) rho = __builtin_nan("");  // (a tile a quarter turn wide: no rule applies)
    cone.cos_rho = cos(rho), cone.sin_rho = sin(rho);
    cone.cos_w = cos(angle), cone.sin_w = sin(angle);

    const uint32_t n_elem = P.n_spheres + P.n_elem_tris;
    uint32_t word = 0;
    bool all = n_elem <= 24u && P.n_meshes <= 7u && P.n_elem_tris == 0u;
    for (uint32_t e = 0; e < n_elem && e < 24u; ++e) {
        const uint32_t desc = P.elems ? P.elems[e] : e;
        bool out = false;
        if (!(desc >> 31)) {
            const DevSphere sp = P.spheres[desc];
            const D3 q = d3(sp.center) - pos;
            const double r = double(sp.radius), l = dlen(q);
            if (r > 0.0 && r < 1e30 && l < 1e30) {  // (false for NaN)
                // (the second term is the analysed one, 15x the bound in the header; the others are loose change on top)
                const double slack = 1e-5 * r + 1e-5 * (l * l + r * r) / r + 1e-6 * l;
                const double reach = r + slack + angle * (l + r);
                const D3 qhat = (1.0 / l) * q;
                out = cols.outside(q, reach) || rows.outside(q, reach) ||
                      (cone.misses(qhat, l, r + slack, 1.0) && (cone.misses(qhat, l, r + slack, -1.0) || cone.backward_deep(qhat, l, r - slack)));
            }
        }
        if (out) word |= 1u << e;
        else all = false;
    }
    // can a forward camera ray of the tile enter this box? (bounding sphere against the cone; corners against the slabs)
    const auto box_unreachable = [&](D3 lo, D3 hi, double slack, double far) -> bool {
        const D3 q = 0.5 * (lo + hi) - pos;
        const double l = dlen(q);
        if (cone.misses((1.0 / l) * q, l, 0.5 * dlen(hi - lo) * 1.001 + slack, 1.0)) return true;
        const double reach = slack + angle * far;
        for (uint32_t slab = 0; slab < 2u; ++slab) {
            const CullSlab& sl = slab ? rows : cols;
            bool pos_side = true, neg_side = true;
            for (uint32_t k = 0; k < 8u; ++k) {
                const D3 v = D3{k & 1u ? hi.x : lo.x, k & 2u ? hi.y : lo.y, k & 4u ? hi.z : lo.z} - pos;
                const double g0 = ddot(sl.n_lo, v), g1 = ddot(sl.n_hi, v);
                pos_side = pos_side && g0 > reach && g1 > reach;
                neg_side = neg_side && g0 < -reach && g1 < -reach;
            }
            if (pos_side || neg_side) return true;
        }
        return false;
    };
    for (uint32_t m = 0; m < P.n_meshes && m < 7u; ++m) {
        const DevMesh& md = P.meshes[m];
        double far = 0.0;
        bool finite = true;
        for (uint32_t k = 0; k < 8u; ++k) {
            const D3 q = D3{double(k & 1u ? md.bbox_hi[0] : md.bbox_lo[0]), double(k & 2u ? md.bbox_hi[1] : md.bbox_lo[1]),
                            double(k & 4u ? md.bbox_hi[2] : md.bbox_lo[2])} - pos;
            const double l = dlen(q);
            finite = finite && l < 1e30;
            far = l > far ? l : far;
        }
        // The mesh's own box, aabbox.rs:28-58 in float (u = 2^-24). Per axis the two quotients t = (plane - o) / d carry one
        // rounding of the difference and one of the division: t~ = t (1 + e), |e| <= 2u + u^2. If the float test passes
        // (t_max >= 0, t_min <= t_max) there is an s >= 0 with, on every axis, min(t~lo, t~hi) <= s <= max(t~lo, t~hi); the
        // point o + s d of the float ray then has every coordinate within 2u |plane - o| + u |plane - o| <= 3u far of the
        // interval [lo, hi] (o + t d = plane exactly for the exact t): it lies within 3 sqrt(3) u far = 3.1e-7 far of the
        // box, on the forward half of the ray. (A zero direction component makes the quotients +-inf or NaN; fmin / fmax
        // skip a NaN, and an axis whose line runs outside its slab gives t_min = +inf or t_max = -inf: a miss, as in
        // geometry.) The slack is 16 u far, three times that bound; the angle between the float ray and the exact
        // line of its pixel is added on top by box_unreachable (angle * far). Round 3 had a chosen 1e-4 far here.
        const double slack = 16.0 * (1.0 / 16777216.0) * far;
        const bool out = finite && box_unreachable(d3(md.bbox_lo), d3(md.bbox_hi), slack, far);
        if (out) word |= 1u << (24u + m);
        // A ray may pass the mesh's box and still have no triangle to hit: the tile is free of the mesh as well when
        // every box in the top kCullLevels levels of its tree is out of the tile's reach. Depth first, descending only where
        // a box is in reach. "Reach" for a tree box is what the TRAVERSAL grows it by (the header of this file's BVH section,
        // make_cull): a triangle the float Moller-Trumbore test accepts has its point within 18 u (K |d| + 1) S of the exact
        // triangle, K = |e1||e2| / eps -- the |a| >= eps rule bounds the amplification, and for a large triangle met at a
        // grazing angle that is far more than a few ulp -- so every child box is grown by the pad the trace kernel itself
        // would use for a ray from this position, 64 u (S + |o|_inf) (max|e1||e2| of the child's subtree * |d| / eps + 1)
        // with |d| <= 1.001: a camera ray that takes a triangle passes inside every ancestor's grown box, so a tile none
        // of whose rays enters a grown box of some level takes no triangle below it.
        const D3 mc = d3(md.center) - pos;
        const double o_inf = fmax(fmax(fabs(pos.x), fabs(pos.y)), fabs(pos.z));
        const double pad_base = 64.0 * (1.0 / 16777216.0) * (dlen(mc) + double(md.radius) + o_inf);
        bool mesh_free = out;
        if (!out && all && finite && md.n_nodes != 0u) {  // (only where the answer matters: nothing else is in reach so far)
            constexpr uint32_t kCullLevels = 6;
            uint32_t stack[3 * kCullLevels + 4];
            uint32_t sp = 0;
            stack[sp++] = 0u;  // (node index << 3 | level)
            mesh_free = true;
            while (sp != 0u && mesh_free) {  // (uniform over the tile's four lanes: they hold the same stack)
                const uint32_t e = stack[--sp];
                const BvhNode4& nd = md.nodes[e >> 3];
                // this lane's child: 0 = no box or out of reach, 1 = in reach and to be opened, 2 = in reach and not to be
                // opened (a leaf, the level limit, a full stack, a box this rule does not understand): the mesh stays
                const uint32_t c = quad_lane;
                const int32_t link = nd.child[c];
                uint32_t state = 0u;
                if (link != kNoChild) {
                    const D3 lo = D3{double(nd.lo_x[c]), double(nd.lo_y[c]), double(nd.lo_z[c])};
                    const D3 hi = D3{double(nd.hi_x[c]), double(nd.hi_y[c]), double(nd.hi_z[c])};
                    if (!(dlen(lo - pos) < 1e30 && dlen(hi - pos) < 1e30)) state = 2u;
                    else if (!box_unreachable(lo, hi, pad_base * (double(nd.max_e12[c]) * 1.001 * double(P.eps_frac) + 1.0), far))
                        state = (link >= 0 && (e & 7u) + 1u < kCullLevels) ? 1u : 2u;
                }
                const uint32_t open = uint32_t(__builtin_amdgcn_ballot_w64(state == 1u) >> quad_shift) & 15u;
                const uint32_t stay = uint32_t(__builtin_amdgcn_ballot_w64(state == 2u) >> quad_shift) & 15u;
                if (stay != 0u) mesh_free = false;
                for (uint32_t k = 0; k < 4u && mesh_free; ++k) {
                    if (!((open >> k) & 1u)) continue;
                    if (sp < 3u * kCullLevels + 4u) stack[sp++] = (uint32_t(nd.child[k]) << 3) | ((e & 7u) + 1u);
                    else mesh_free = false;
                }
            }
        }
        if (!mesh_free) all = false;
    }
    if (all) word |= 1u << 31;
    if (quad_lane == 0u) P.tile_cull[tile] = word;
}

// The rank's tiles split by what primary_cull_kernel found, each list in ascending order (TraceParams::tile_lists): one
// workgroup, every thread a contiguous run of local tiles, an exclusive scan of the runs' counts in between.
// (ONE wave and no LDS -- the scan is made of wave shuffles: when the camera changes in the middle of a stream of frames this
// kernel has to find room beside resident trace waves, which hold every wave slot and all of every CU's LDS. A 1024-thread
// workgroup waited milliseconds for a whole free CU (round 3), the 256-thread one with its 2 KB of LDS that followed still
// needed four free slots on one CU at once (round 4's trace: up to 3 ms).)
// kListBlock = 256 (four waves, 32 bytes of LDS for their totals) is the form for a launch that has the GPU to itself -- nothing
// to fit beside, and a quarter of the table per wave: 0.075 -> 0.03 ms in front of a blocking frame of a new camera.
template <int kListBlock>
__global__ __launch_bounds__(kListBlock) void tile_lists_kernel(const TraceParams P) {
    const uint32_t n = P.n_local_tiles, t = threadIdx.x;
    const uint32_t per = (n + kListBlock - 1) / kListBlock;
    const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    // class of a tile: 2 = background only; 1 = light (no mesh box in reach, at most one sphere); 0 = heavy
    const uint32_t el_mask = (1u << (P.n_spheres < 24u ? P.n_spheres : 24u)) - 1u, me_mask = (1u << (P.n_meshes < 7u ? P.n_meshes : 7u)) - 1u;
    const auto class_of = [&](uint32_t word) -> uint32_t {
        if (word >> 31) return 2u;
        if (P.tile_list_mode == 0u) return 0u;
        // (the table's element bits are indexed by position in Scene::elements and are only ever set for spheres: with
        // BasicTriangle elements in the scene, which are never culled, no tile is called light)
        const bool light = ((word >> 24) & me_mask) == me_mask && __popc(~word & el_mask) <= 1 && P.n_spheres <= 24u && P.n_meshes <= 7u &&
                           P.n_elem_tris == 0u;
        return light ? 1u : 0u;
    };
    // (one wave walks the whole table, every lane a run of it: the table's words are fetched eight at a time, so that a
    // lane's loads are in flight together instead of one dependent round trip per tile)
    constexpr uint32_t kBatch = 8;
    const auto load_batch = [&](uint32_t tl0, uint32_t (&w)[kBatch]) {
#pragma unroll
        for (uint32_t j = 0; j < kBatch; ++j) w[j] = tl0 + j < hi ? P.tile_cull[(tl0 + j) * P.tile_world + P.tile_rank] : 0x80000000u;
    };
    uint32_t mine[2] = {0u, 0u};  // heavy / light tiles of this run (both go to the trace kernel)
    for (uint32_t tl0 = lo; tl0 < hi; tl0 += kBatch) {
        uint32_t w[kBatch];
        load_batch(tl0, w);
#pragma unroll
        for (uint32_t j = 0; j < kBatch; ++j) {
            const uint32_t c = class_of(w[j]);
            if (tl0 + j < hi && c < 2u) ++mine[c];
        }
    }
    uint32_t incl[2] = {mine[0], mine[1]};
    const uint32_t wlane = t & 63u;
    for (uint32_t d = 1; d < 64u; d <<= 1) {  // inclusive scans over the wave
        const uint32_t v0 = uint32_t(__shfl_up(int(incl[0]), d, 64)), v1 = uint32_t(__shfl_up(int(incl[1]), d, 64));
        if (wlane >= d) incl[0] += v0, incl[1] += v1;
    }
    uint32_t n_heavy = uint32_t(__shfl(int(incl[0]), 63, 64)), n_light = uint32_t(__shfl(int(incl[1]), 63, 64));
    if (kListBlock > 64) {  // ... and over the workgroup's waves, through their totals
        __shared__ uint32_t wave_total[2][kListBlock / 64];
        if (wlane == 63u) wave_total[0][t >> 6] = n_heavy, wave_total[1][t >> 6] = n_light;
        __syncthreads();
        n_heavy = n_light = 0u;
        for (uint32_t w = 0; w < uint32_t(kListBlock / 64); ++w) {
            if (w < (t >> 6)) incl[0] += wave_total[0][w], incl[1] += wave_total[1][w];
            n_heavy += wave_total[0][w], n_light += wave_total[1][w];
        }
    }
    const uint32_t n_work = n_heavy + n_light;
    uint32_t r[2] = {incl[0] - mine[0], incl[1] - mine[1]};  // heavy / light tiles before this run
    uint32_t k = lo - r[0] - r[1];                                  // background-only tiles before it
    uint32_t* const work = P.tile_lists + kTileListHeader;
    uint32_t* const sky = work + n;
    // where the classes go in the work list (tile_list_mode; 0: one class, ascending): 1 heavy | light, 2 light/2 | heavy |
    // light/2, 3 light | heavy -- each class ascending in itself; 4: the image's row-major order in the direction it is handed
    // out in (P.tiles_reversed: from the last tile down), except that the LAST n_work / tile_tail_div light tiles of that
    // order are taken out of it and handed out at the very end: the bulk of the launch keeps row-major order's mix of
    // traversal-heavy and shading-heavy work in every wave, and the launch drains on light tiles whatever the image
    const uint32_t half = n_light / 2u;
    uint32_t n_tail = n_work / (P.tile_tail_div != 0u ? P.tile_tail_div : 8u);
    n_tail = n_tail < n_light ? n_tail : n_light;
    const uint32_t n_front_light = n_light - n_tail;  // light tiles that stay in row-major order
    for (uint32_t tl0 = lo; tl0 < hi; tl0 += kBatch) {
      uint32_t w[kBatch];
      load_batch(tl0, w);
#pragma unroll
      for (uint32_t j = 0; j < kBatch; ++j) {
        const uint32_t tl = tl0 + j;
        if (tl >= hi) continue;
        const uint32_t c = class_of(w[j]);
        if (c == 2u) {
            sky[k++] = tl;
            continue;
        }
        uint32_t pos;
        if (P.tile_list_mode == 4u) {
            // ranks in hand-out order: heavy and light tiles handed out BEFORE this one
            const bool rev = P.tiles_reversed != 0u;
            const uint32_t h = c == 0u ? (rev ? n_heavy - 1u - r[0] : r[0]) : (rev ? n_heavy - r[0] : r[0]);
            const uint32_t l = c == 1u ? (rev ? n_light - 1u - r[1] : r[1]) : (rev ? n_light - r[1] : r[1]);
            if (c == 1u && l >= n_front_light) pos = (n_work - n_tail) + (l - n_front_light);
            else pos = h + (l < n_front_light ? l : n_front_light);
            ++r[c];
        } else if (c == 0u) {
            pos = (P.tile_list_mode == 2u ? half : P.tile_list_mode == 3u ? n_light : 0u) + r[0]++;
        } else {
            pos = (P.tile_list_mode == 1u ? n_heavy : P.tile_list_mode == 2u && r[1] >= half ? n_heavy : 0u) + r[1]++;
        }
        work[pos] = tl;
      }
    }
    if (t == kListBlock - 1) {
        P.tile_lists[0] = n_work;
        P.tile_lists[1] = n - n_work;
        P.tile_lists[2] = n_light, P.tile_lists[3] = 0u;
    }
}

__host__ __device__ inline uint32_t local_tiles_of(uint32_t n_tiles, uint32_t rank, uint32_t world) {
    return rank < n_tiles ? (n_tiles - rank + world - 1u) / world : 0u;
}

// Gathered per-rank packed tiles -> row-major image (rank r's block starts after the blocks of
// ranks < r; inside a block tiles are in ascending global tile order).
__global__ __launch_bounds__(kSmallBlock) void unpack_kernel(const float* __restrict__ gathered, uint32_t width,
                                                        uint32_t height, uint32_t tiles_x, uint32_t n_tiles,
                                                        uint32_t world, size_t rank_stride_pixels, float* out_radiance,
                                                        uint8_t* out_rgb8) {
    const size_t i = size_t(blockIdx.x) * kSmallBlock + threadIdx.x;
    if (i >= size_t(width) * height) return;
    const uint32_t row = uint32_t(i / width), col = uint32_t(i - size_t(row) * width);
    const uint32_t tile = tile_number(row / RBRT_TILE, col / RBRT_TILE, tiles_x);
    const uint32_t rank = tile % world, tile_local = tile / world;
    size_t base = 0;
    if (rank_stride_pixels != 0) {  // equal-size slots per rank (what a gather of equal-size tensors produces)
        base = size_t(rank) * rank_stride_pixels;
    } else {                        // tightly packed
        for (uint32_t r = 0; r < rank; ++r) base += size_t(local_tiles_of(n_tiles, r, world)) * 64u;
    }
    const size_t src = (base + size_t(tile_local) * 64u + (row % RBRT_TILE) * RBRT_TILE + col % RBRT_TILE) * 3u;
    const float x = gathered[src], y = gathered[src + 1], z = gathered[src + 2];
    if (out_radiance) {
        out_radiance[i * 3 + 0] = x;
        out_radiance[i * 3 + 1] = y;
        out_radiance[i * 3 + 2] = z;
    }
    if (out_rgb8) {
        out_rgb8[i * 3 + 0] = quantise(x);
        out_rgb8[i * 3 + 1] = quantise(y);
        out_rgb8[i * 3 + 2] = quantise(z);
    }
}

// Scene::hit for arbitrary rays (test / diagnostic hook behind rbrt_hip_trace_rays).
constexpr int kRaysBlock = 128;
__global__ __launch_bounds__(kRaysBlock) void trace_rays_kernel(const TraceParams P, const float* __restrict__ rays,
                                                            size_t n, float* out_t, int32_t* out_obj,
                                                            int32_t* out_tri, float* out_dist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* stack = lds + threadIdx.x;
    const size_t i = size_t(blockIdx.x) * kRaysBlock + threadIdx.x;
    if (i >= n) return;
    const V3 o = mk(rays + 6 * i), d = mk(rays + 6 * i + 3);
    HitRec h;
    LocalCounters lc = {0, 0, 0, 0, 0};
    scene_hit<false>(P, o, d, stack, uint32_t(kRaysBlock), h, lc);
    const bool hit = h.obj >= 0;
    const float nanv = __int_as_float(0x7fc00000);
    if (out_t) out_t[i] = hit ? h.t : nanv;
    if (out_obj) out_obj[i] = h.obj;
    if (out_tri) out_tri[i] = (hit && uint32_t(h.obj) >= P.n_spheres + P.n_elem_tris) ? int32_t(h.tri) : -1;
    if (out_dist) out_dist[i] = hit ? h.dist : nanv;
}

// Test hook (rbrt_hip_selftest_ieee): the short IEEE forms (sqrt_core inside ieee_sqrt, normalize) against the
// compiler's on pseudo-random operands. Waves 0,1,2 (mod 4) draw every lane's operands from the in-range domain, so
// the short forms are what runs; waves 3 (mod 4) mix in zeros, denormals, huge, inf and NaN: the ballot sends those
// waves through the compiler's forms, which must then be what comes out. counts[0] = mismatching sqrt results,
// counts[1] = mismatching normalize components, counts[2] = lanes that took the short path (in-range waves).
__global__ __launch_bounds__(kBlock) void ieee_selftest_kernel(uint64_t seed, size_t n, unsigned long long* counts) {
    const size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    Rng rng;
    rng.init(seed, uint32_t(i >> 20), uint32_t(i));
    const uint32_t mode = uint32_t(i >> 6) & 3u;
    auto draw = [&](int e_lo, int e_hi) {  // +-(1.m) * 2^e, e uniform in [e_lo, e_hi]
        const uint32_t r = rng.next_u32(), q = rng.next_u32();
        const int e = e_lo + int(q % uint32_t(e_hi - e_lo + 1));
        return __uint_as_float((r & 0x807FFFFFu) | (uint32_t(e + 127) << 23));
    };
    V3 a;
    float x;
    if (mode != 3u) {
        a = mk(draw(-39, 49), draw(-39, 49), draw(-39, 49));
        x = __builtin_fabsf(draw(-79, 99));
    } else {
        const uint32_t pick = rng.next_u32();
        a = mk(draw(-126, 127), draw(-126, 127), draw(-126, 127));
        x = __builtin_fabsf(draw(-126, 127));
        const float specials[8] = {0.0f, -0.0f, __uint_as_float(1u), __uint_as_float(0x007FFFFFu), __uint_as_float(0x7F7FFFFFu),
                                   __uint_as_float(0x7F800000u), __uint_as_float(0xFF800000u), __uint_as_float(0x7FC00000u)};
        if ((pick & 3u) == 0u) a.x = specials[(pick >> 2) & 7u];
        if ((pick & 12u) == 0u) a.y = specials[(pick >> 5) & 7u];
        if ((pick & 48u) == 0u) x = __builtin_fabsf(specials[(pick >> 8) & 7u]);
    }
    auto same = [](float p, float q) { return __float_as_uint(p) == __float_as_uint(q) || (p != p && q != q); };
    const float s_fast = ieee_sqrt(x), s_ref = __builtin_sqrtf(x);
    const V3 n_fast = normalize(a), n_ref = normalize_ieee(a);
    unsigned long long bad_s = same(s_fast, s_ref) ? 0ull : 1ull;
    unsigned long long bad_n = (same(n_fast.x, n_ref.x) ? 0ull : 1ull) + (same(n_fast.y, n_ref.y) ? 0ull : 1ull) + (same(n_fast.z, n_ref.z) ? 0ull : 1ull);
    if (bad_s) atomicAdd(&counts[0], bad_s);
    if (bad_n) atomicAdd(&counts[1], bad_n);
    if (mode != 3u && threadIdx.x % 64 == 0) atomicAdd(&counts[2], 64ull);
}

// Test hook (rbrt_hip_selftest_gate): both forms of the mesh gate on arbitrary rays against one box.
__global__ __launch_bounds__(kBlock) void gate_selftest_kernel(const float* __restrict__ box, const float* __restrict__ rays, size_t n,
                                                               uint8_t* out_fast, uint8_t* out_exact) {
    const size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const V3 o = mk(rays + 6 * i), d = mk(rays + 6 * i + 3);
    out_fast[i] = bbox_gate_fast(box, box + 3, o, d) ? 1 : 0;
    out_exact[i] = bbox_gate(box, box + 3, o, d) ? 1 : 0;
}

// Test hook (rbrt_hip_debug_scatter): one RayScattering::scatter event per thread, through the same device functions
// the megakernel's shading passes are built from (scatter / reflect / refract / schlick / random_point_in_unit_sphere).
__global__ __launch_bounds__(kBlock) void scatter_debug_kernel(const DevMaterial* __restrict__ mats, const float* __restrict__ in_dir,
                                                                const float* __restrict__ p, const float* __restrict__ normal,
                                                                const uint32_t* __restrict__ rng_state, size_t n, float* out_dir,
                                                                uint8_t* out_ok, uint32_t* out_rng_state) {
    const size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    Rng rng = {rng_state[2 * i], rng_state[2 * i + 1]};
    V3 nd = mk(0.0f, 0.0f, 0.0f);
    const bool ok = scatter(mats[i], mk(in_dir + 3 * i), mk(p + 3 * i), mk(normal + 3 * i), rng, nd);
    if (out_dir) out_dir[3 * i] = nd.x, out_dir[3 * i + 1] = nd.y, out_dir[3 * i + 2] = nd.z;
    if (out_ok) out_ok[i] = ok ? 1 : 0;
    if (out_rng_state) out_rng_state[2 * i] = rng.s0, out_rng_state[2 * i + 1] = rng.s1;
}

// ---------------------------------------------------------------------------------------------
// Launch wrappers (called from api.cpp, which is plain C++)
// ---------------------------------------------------------------------------------------------
size_t megakernel_gseq_bytes(uint32_t n_waves, uint32_t pool) { return size_t(n_waves) * pool * kSeqWords * sizeof(uint32_t); }
size_t megakernel_gstack_bytes(uint32_t n_waves) { return size_t(n_waves) * kStackMax * 64u * sizeof(uint32_t); }

__host__ __device__ inline uint32_t megakernel_lds_dwords(uint32_t pool, uint32_t stack_entries, uint32_t n_spheres, uint32_t n_meshes,
                                                          uint32_t n_elem_tris) {
    const uint32_t n_elem = n_spheres + n_elem_tris;
    const uint32_t scene = n_spheres * kSphDw + (n_elem + n_meshes) * kMatDw + n_meshes * kMeshDw + kGenDw +
                           (n_elem_tris != 0u ? n_elem_tris * kTriDw + n_elem : 0u);  // triangle table + element order
    const uint32_t pool_pad = (pool + 63u) & ~63u;  // status + list: one byte per (padded) slot each
    uint32_t dw = uint32_t(kFields) * pool + kCellDw + kTqDw + kHelpDw + pool_pad / 2u + stack_entries * 64u + scene;
    if (RBRT_REGION_TIMERS) dw += uint32_t(kNumRegions);  // analysis build: a u32 cycle accumulator per region
    return dw;
}
size_t megakernel_lds_bytes(uint32_t pool, uint32_t stack_entries, uint32_t n_spheres, uint32_t n_meshes, uint32_t n_elem_tris) {
    return size_t(megakernel_lds_dwords(pool, stack_entries, n_spheres, n_meshes, n_elem_tris)) * sizeof(uint32_t);
}

// What the HIP runtime says fits: resident single-wave workgroups of the trace kernel per CU at this much LDS (0 on error).
int megakernel_occupancy_per_cu(uint32_t pool, size_t lds_bytes) {
    int n = 0;
    hipError_t e = pool == 256 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_megakernel<256, false, true>, 64, lds_bytes)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_megakernel<128, false, true>, 64, lds_bytes);
    return e == hipSuccess ? n : 0;
}

// n_waves single-wave workgroups; each loops until the global work counter (zeroed by the caller on
// this stream) runs out, so any grid size is correct and no wave ever waits on another.
// A helper launch (api.cpp "Elastic launches"): more waves for a launch that is already running -- the same parameters
// (work counters, sample buffer, tables), scratch slots from P.wave_base on.
hipError_t launch_trace_helper(const TraceParams& P, uint32_t n_waves, uint32_t pool, bool share, hipStream_t stream) {
    if (P.n_items == 0 || n_waves == 0 || !P.helper_words) return hipSuccess;
    const size_t lds = megakernel_lds_bytes(pool, P.stack_entries, P.n_spheres, P.n_meshes, P.n_elem_tris);
    if (pool == 128 && share)
        hipLaunchKernelGGL((trace_megakernel<128, false, true, true>), dim3(n_waves), dim3(64), lds, stream, P);
    else if (pool == 128)
        hipLaunchKernelGGL((trace_megakernel<128, false, false, true>), dim3(n_waves), dim3(64), lds, stream, P);
    else
        return hipErrorInvalidValue;  // (the lab's 256-slot pool has no helper build)
    return hipGetLastError();
}

hipError_t launch_trace_megakernel(const TraceParams& P, uint32_t n_waves, uint32_t pool, bool stats, bool share,
                                   hipStream_t stream) {
    if (P.n_items == 0 || n_waves == 0) return hipSuccess;
    const size_t lds = megakernel_lds_bytes(pool, P.stack_entries, P.n_spheres, P.n_meshes, P.n_elem_tris);
#define RBRT_LAUNCH_MK(POOLN, STATS, SHARE) \
    hipLaunchKernelGGL((trace_megakernel<POOLN, STATS, SHARE>), dim3(n_waves), dim3(64), lds, stream, P)
#define RBRT_LAUNCH_POOL(POOLN)                            \
    do {                                                   \
        if (stats && share) RBRT_LAUNCH_MK(POOLN, true, true);    \
        else if (stats) RBRT_LAUNCH_MK(POOLN, true, false);       \
        else if (share) RBRT_LAUNCH_MK(POOLN, false, true);       \
        else RBRT_LAUNCH_MK(POOLN, false, false);                 \
    } while (0)
    if (pool == 128)
        RBRT_LAUNCH_POOL(128);
    else if (pool == 256)
        RBRT_LAUNCH_POOL(256);
    else
        return hipErrorInvalidValue;
#undef RBRT_LAUNCH_POOL
#undef RBRT_LAUNCH_MK
    return hipGetLastError();
}

// The tile pass: the culling table of P.cam, then (tile_lists given) the rank's two tile lists from it.
hipError_t launch_primary_cull(const TraceParams& P, hipStream_t stream) {
    if (P.n_tiles == 0 || !P.tile_cull) return hipSuccess;
    const uint32_t cull_tiles = P.tile_world > 1u ? local_tiles_of(P.n_tiles, P.tile_rank, P.tile_world) : P.n_tiles;
    if (cull_tiles == 0u) return hipSuccess;
    const size_t cull_threads = size_t(cull_tiles) * kCullLanes;
    hipLaunchKernelGGL(primary_cull_kernel, dim3(uint32_t((cull_threads + kCullBlock - 1) / kCullBlock)), dim3(kCullBlock), 0, stream, P);
    if (P.tile_lists) {
        if (P.tile_lists_wide) hipLaunchKernelGGL(tile_lists_kernel<256>, dim3(1), dim3(256), 0, stream, P);
        else hipLaunchKernelGGL(tile_lists_kernel<64>, dim3(1), dim3(64), 0, stream, P);
    }
    return hipGetLastError();
}

// One empty launch: the HIP runtime loads a library's device code at the first launch of one of its kernels (7-8 ms for this
// one). rbrt_hip_scene_create does it, so that the first render of a process is a render.
__global__ void code_load_kernel() {}
hipError_t launch_code_load(hipStream_t stream) {
    hipLaunchKernelGGL(code_load_kernel, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

hipError_t launch_sky_resolve(const TraceParams& P, const ResolveParams& R, hipStream_t stream) {
    const size_t npix = size_t(R.n_local_tiles) * 64u;  // (an upper bound: the list's length is known on the device)
    if (npix == 0 || !R.tile_lists) return hipSuccess;
    hipLaunchKernelGGL(sky_resolve_kernel, dim3(uint32_t((npix + kSmallBlock - 1) / kSmallBlock)), dim3(kSmallBlock), 0, stream, P, R);
    return hipGetLastError();
}

hipError_t launch_resolve(const ResolveParams& R, hipStream_t stream) {
    const size_t npix = size_t(R.n_local_tiles) * 64u;
    if (npix == 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_kernel, dim3(uint32_t((npix + kSmallBlock - 1) / kSmallBlock)), dim3(kSmallBlock), 0, stream, R);
    return hipGetLastError();
}

hipError_t launch_unpack(const float* gathered, uint32_t width, uint32_t height, uint32_t world,
                         size_t rank_stride_pixels, float* out_radiance, uint8_t* out_rgb8, hipStream_t stream) {
    const uint32_t tiles_x = (width + RBRT_TILE - 1) / RBRT_TILE, tiles_y = (height + RBRT_TILE - 1) / RBRT_TILE;
    const size_t n = size_t(width) * height;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_kernel, dim3(uint32_t((n + kSmallBlock - 1) / kSmallBlock)), dim3(kSmallBlock), 0, stream, gathered,
                       width, height, tiles_x, tiles_x * tiles_y, world, rank_stride_pixels, out_radiance, out_rgb8);
    return hipGetLastError();
}

hipError_t launch_trace_rays(const TraceParams& P, const float* rays, size_t n, float* out_t, int32_t* out_obj,
                             int32_t* out_tri, float* out_dist, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(trace_rays_kernel, dim3(uint32_t((n + kRaysBlock - 1) / kRaysBlock)), dim3(kRaysBlock),
                       size_t(kStackMax) * kRaysBlock * sizeof(uint32_t), stream, P, rays, n, out_t, out_obj,
                       out_tri, out_dist);
    return hipGetLastError();
}

hipError_t launch_gate_selftest(const float* d_box, const float* d_rays, size_t n, uint8_t* d_fast, uint8_t* d_exact) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gate_selftest_kernel, dim3(uint32_t((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, d_box, d_rays, n,
                       d_fast, d_exact);
    return hipGetLastError();
}

hipError_t launch_ieee_selftest(uint64_t seed, size_t n, unsigned long long* d_counts) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(ieee_selftest_kernel, dim3(uint32_t((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, seed, n, d_counts);
    return hipGetLastError();
}
hipError_t launch_scatter_debug(const DevMaterial* d_mats, const float* d_in_dir, const float* d_p, const float* d_normal,
                                 const uint32_t* d_rng, size_t n, float* d_out_dir, uint8_t* d_out_ok, uint32_t* d_out_rng) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_debug_kernel, dim3(uint32_t((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, d_mats, d_in_dir, d_p,
                       d_normal, d_rng, n, d_out_dir, d_out_ok, d_out_rng);
    return hipGetLastError();
}

uint64_t host_splitmix64(uint64_t x) { return splitmix64(x); }

}  // namespace rbrt
