// rbrt_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference's AVX render path (baurst/rbrt, Rust), used only as
// the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
// Nothing under rbrt_amd/ may link, load or call this file.
//
// The reference itself cannot be built here (Rust; no cargo/rustc in the image), so this is a
// "port"-kind oracle. It is pinned by every known-answer test the reference's own unit tests
// hold for this path (tests/test_oracle_kats.py lists them with their file:line). Two things are
// unpinnable and said so in DESIGN.md: the random stream (the reference draws from an OS-seeded
// thread-local generator: cam.rs:69,71; materials.rs:17-19,24-26; dielectric.rs:48) and
// everything no reference test touches (triangle intersection, bbox gate, camera, integrator).
//
// Build: g++ -O2 -mavx -ffp-contract=off (no -mfma, no fast-math) — see oracle/Makefile.
// Arithmetic rules kept from the reference: f32 everywhere, mul then add (no FMA: rustc never
// contracts; vec3_avx.rs:18-21,40-42 use separate mul/add), dot = (x*x' + y*y') + z*z'
// (vec3.rs:115-117,157-159), normalize = three divisions (vec3.rs:119-126).

#include <immintrin.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rbrt_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Vec3 / Ray  (vec3.rs:5-160, ray.rs:3-19)
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }  // vec3.rs:23-34
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }  // vec3.rs:12-22
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }  // vec3.rs:55-65
inline V3 operator*(float s, V3 b) { return V3{s * b.x, s * b.y, s * b.z}; }     // vec3.rs:66-76
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }     // vec3.rs:78-88
inline float sum(V3 a) { return a.x + a.y + a.z; }                              // vec3.rs:115-117
inline float dot(V3 a, V3 b) { return sum(a * b); }                             // vec3.rs:157-159
inline float length(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // vec3.rs:111-113
inline V3 normalize(V3 a) {                                                     // vec3.rs:119-126
    float len = length(a);
    return V3{a.x / len, a.y / len, a.z / len};
}
inline V3 cross(V3 a, V3 b) {  // vec3.rs:128-134
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// vec3.rs:139-155, Z-X-Z Euler; f32 sin/cos are libm sinf/cosf like Rust's on linux-gnu.
inline V3 rotate_point(V3 p, V3 rot) {
    float s_x = std::sin(rot.x), s_y = std::sin(rot.y), s_z = std::sin(rot.z);
    float c_x = std::cos(rot.x), c_y = std::cos(rot.y), c_z = std::cos(rot.z);
    float x = p.x, y = p.y, z = p.z;
    return V3{(c_x * c_z - c_y * s_x * s_z) * x - (c_x * s_z + c_y * c_z * s_x) * y + s_x * s_y * z,
              (c_z * s_x + c_x * c_y * s_z) * x + (c_x * c_y * c_z - s_x * s_z) * y - c_x * s_y * z,
              s_y * s_z * x + c_z * s_y * y + c_y * z};
}

struct Ray {
    V3 origin, direction;
    V3 point_at(float t) const { return origin + t * direction; }  // ray.rs:10-12
};

// ---------------------------------------------------------------------------------------------
// The build's seeded random stream (DESIGN.md "RNG"). The reference has none to restate: this is
// the contract both this oracle and the HIP kernel implement independently.
//   key   = splitmix64( splitmix64(seed) ^ (pixel << 32 | sample) ),  pixel = row*W + col
//   state = (s0, s1) = (lo32(key), hi32(key)), (0,0) replaced by (1,0)
//   draw  = xoroshiro64** 1.0; f32 = (u32 >> 8) * 2^-24  (rand 0.8 `Standard` mapping for f32)
// ---------------------------------------------------------------------------------------------
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

struct Rng {
    uint32_t s0, s1;
    Rng(uint64_t seed, uint32_t pixel, uint32_t sample) {
        uint64_t key = splitmix64(splitmix64(seed) ^ ((uint64_t(pixel) << 32) | uint64_t(sample)));
        s0 = uint32_t(key);
        s1 = uint32_t(key >> 32);
        if (s0 == 0 && s1 == 0) s0 = 1;
    }
    uint32_t next_u32() {
        uint32_t r = rotl32(s0 * 0x9E3779BBu, 5) * 5u;
        uint32_t t = s1 ^ s0;
        s0 = rotl32(s0, 26) ^ t ^ (t << 9);
        s1 = rotl32(t, 13);
        return r;
    }
    float next_f32() { return float(next_u32() >> 8) * (1.0f / 16777216.0f); }
};

// ---------------------------------------------------------------------------------------------
// Materials  (materials.rs, lambertian.rs, metal.rs, dielectric.rs)
// ---------------------------------------------------------------------------------------------
struct Hit {
    V3 point, normal;
    const rbrt_material_t* mat;
    float dist;
};

// materials.rs:14-30 — draws x, y, z in that order; accept when length <= 1.0
inline V3 random_point_in_unit_sphere(Rng& rng) {
    float x = rng.next_f32(), y = rng.next_f32(), z = rng.next_f32();
    V3 p = 2.0f * v3(x, y, z) - v3(1.0f, 1.0f, 1.0f);
    while (length(p) > 1.0f) {
        x = rng.next_f32();
        y = rng.next_f32();
        z = rng.next_f32();
        p = 2.0f * v3(x, y, z) - v3(1.0f, 1.0f, 1.0f);
    }
    return p;
}

// materials.rs:32-37
inline V3 reflect(V3 dir, V3 n) {
    V3 d = normalize(dir);
    V3 nu = normalize(n);
    V3 r = d - 2.0f * nu * dot(d, nu);
    return normalize(r);
}

// f32::powi(n) with a constant exponent is LLVM's multiply chain (square-and-multiply, LSB first);
// for 2: x*x, for 5: x * ((x*x)*(x*x)).
inline float powi2(float x) { return x * x; }
inline float powi5(float x) {
    float x2 = x * x;
    float x4 = x2 * x2;
    return x * x4;
}

// dielectric.rs:63-66
inline float schlick(float cosine, float ref_index) {
    float r0 = powi2((1.0f - ref_index) / (1.0f + ref_index));
    return r0 + (1.0f - r0) * powi5(1.0f - cosine);
}

// dielectric.rs:68-85
inline bool refract(V3 dir, V3 n, float ni_over_nt, V3& out) {
    V3 v = normalize(dir);
    V3 nu = normalize(n);
    float c = dot(v, nu);
    float discr = 1.0f - powi2(ni_over_nt) * (1.0f - powi2(c));
    if (discr > 0.0f) {
        out = ni_over_nt * (v - nu * c) - std::sqrt(discr) * nu;
        return true;
    }
    return false;
}

inline bool scatter(const rbrt_material_t& m, const Ray& in, const Hit& h, V3& att, Ray& out,
                    Rng& rng) {
    switch (m.kind) {
        case RBRT_MAT_LAMBERTIAN: {  // lambertian.rs:11-24
            V3 target = h.point + normalize(h.normal) + random_point_in_unit_sphere(rng);
            out.direction = normalize(target - h.point);
            out.origin = h.point;
            att = v3(m.albedo);
            return true;
        }
        case RBRT_MAT_METAL: {  // metal.rs:12-25
            V3 target = reflect(in.direction, h.normal);
            out.direction = normalize(target + m.param * random_point_in_unit_sphere(rng));
            out.origin = h.point;
            att = v3(m.albedo);
            return dot(out.direction, h.normal) > 0.0f;
        }
        default: {  // dielectric.rs:11-59
            att = v3(1.0f, 1.0f, 1.0f);
            V3 reflected = reflect(in.direction, h.normal);
            V3 outward;
            float ni_over_nt, cosine;
            float a = dot(normalize(in.direction), normalize(h.normal));
            if (a > 0.0f) {
                outward = -1.0f * h.normal;
                ni_over_nt = m.param;
                cosine = m.param * a;
            } else {
                outward = h.normal;
                ni_over_nt = 1.0f / m.param;
                cosine = -a;
            }
            V3 refracted = v3(0, 0, 0);
            float reflect_prob =
                refract(in.direction, outward, ni_over_nt, refracted) ? schlick(cosine, m.param) : 1.0f;
            out.origin = h.point;
            out.direction = (rng.next_f32() < reflect_prob) ? reflected : refracted;
            return true;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Geometry
// ---------------------------------------------------------------------------------------------
std::atomic<uint64_t> g_nan_discriminants{0};

// sphere.rs:20-66. A NaN discriminant panics in the reference (sphere.rs:33); here: miss + count.
inline bool sphere_hit(const rbrt_sphere_t& s, const Ray& ray, float min_dist, float max_dist,
                       Hit& out, float* t_out = nullptr) {
    V3 c = v3(s.center);
    float a = dot(ray.direction, ray.direction);
    V3 l = ray.origin - c;
    float b = dot(ray.direction * 2.0f, l);
    float cc = dot(l, l) - s.radius * s.radius;
    float sol = b * b - 4.0f * a * cc;
    if (sol != sol) {
        g_nan_discriminants.fetch_add(1, std::memory_order_relaxed);
        return false;
    }
    if (sol < 0.0f) return false;
    bool two = sol > 0.0f;
    float t = (-b - std::sqrt(sol)) / (2.0f * a);
    if (two && t < 0.0f) {
        t = (-b + std::sqrt(sol)) / (2.0f * a);
        if (t < 0.0f) return false;
    }
    V3 p = ray.point_at(t);
    float dist = length(ray.origin - p);
    if (dist < min_dist || dist > max_dist) return false;
    out.normal = p - c;
    out.point = p;
    out.mat = &s.mat;
    out.dist = dist;
    if (t_out) *t_out = t;
    return true;
}

// aabbox.rs:11-17: f32::max/min ignore a NaN operand — fmaxf/fminf do the same.
inline float rmax(float a, float b) { return std::fmax(a, b); }
inline float rmin(float a, float b) { return std::fmin(a, b); }

// aabbox.rs:28-58
inline bool bbox_hit(const float lo[3], const float hi[3], const Ray& ray) {
    float t_lower_x = (lo[0] - ray.origin.x) / ray.direction.x;
    float t_upper_x = (hi[0] - ray.origin.x) / ray.direction.x;
    float t_lower_y = (lo[1] - ray.origin.y) / ray.direction.y;
    float t_upper_y = (hi[1] - ray.origin.y) / ray.direction.y;
    float t_lower_z = (lo[2] - ray.origin.z) / ray.direction.z;
    float t_upper_z = (hi[2] - ray.origin.z) / ray.direction.z;
    float t_min_x = rmin(t_lower_x, t_upper_x);
    float t_min_y = rmin(t_lower_y, t_upper_y);
    float t_min_z = rmin(t_lower_z, t_upper_z);
    float t_min = rmax(rmax(t_min_x, t_min_y), t_min_z);
    float t_max_x = rmax(t_lower_x, t_upper_x);
    float t_max_y = rmax(t_lower_y, t_upper_y);
    float t_max_z = rmax(t_lower_z, t_upper_z);
    float t_max = rmin(rmin(t_max_x, t_max_y), t_max_z);
    if (t_max < 0.0f) return false;
    if (t_min > t_max) return false;
    return true;
}

// vec3_avx.rs:10-22
inline __m256 avx_dot(__m256 ax, __m256 ay, __m256 az, __m256 bx, __m256 by, __m256 bz) {
    __m256 px = _mm256_mul_ps(ax, bx);
    __m256 py = _mm256_mul_ps(ay, by);
    __m256 pz = _mm256_mul_ps(az, bz);
    return _mm256_add_ps(_mm256_add_ps(px, py), pz);
}
// vec3_avx.rs:32-45
inline void avx_cross(__m256 ax, __m256 ay, __m256 az, __m256 bx, __m256 by, __m256 bz, __m256& cx,
                      __m256& cy, __m256& cz) {
    cx = _mm256_sub_ps(_mm256_mul_ps(ay, bz), _mm256_mul_ps(az, by));
    cy = _mm256_sub_ps(_mm256_mul_ps(az, bx), _mm256_mul_ps(ax, bz));
    cz = _mm256_sub_ps(_mm256_mul_ps(ax, by), _mm256_mul_ps(ay, bx));
}

// triangle.rs:392-410
inline bool find_smallest_element_bigger_than_eps(const float* ray_params, size_t n,
                                                  const uint8_t* is_padding, float eps,
                                                  float& t_out, size_t& idx_out) {
    size_t min_idx = 0;
    float min_param = 1000000.0f;
    for (size_t idx = 0; idx < n; ++idx) {
        float rp = ray_params[idx];
        if (rp > eps && rp < min_param && !is_padding[idx]) {
            min_param = rp;
            min_idx = idx;
        }
    }
    if (min_param > eps && min_param < 100000.0f) {
        t_out = min_param;
        idx_out = min_idx;
        return true;
    }
    return false;
}

// triangle.rs:134-262 — Moller-Trumbore over ALL triangles, 8 per iteration, results into a
// per-call heap buffer, then the scalar arg-min. chunks_exact(8) drops the tail (triangle.rs:166-167).
// `ray_params` out-param lets the KAT tests look at the per-triangle results.
bool triangle_soa_avx_intersect_with_ray(const Ray& ray, const rbrt_mesh_t& m, float min_dist,
                                         float& t_out, size_t& idx_out,
                                         std::vector<float>* keep_params = nullptr) {
    std::vector<float> ray_params;
    ray_params.reserve(m.n_total);
    const float eps_f32 = min_dist;
    const __m256 eps = _mm256_set1_ps(eps_f32);
    const __m256 eps_frac = _mm256_set1_ps(1.0f / eps_f32);
    const __m256 neg_eps = _mm256_set1_ps(-eps_f32);
    const __m256 zero = _mm256_set1_ps(0.0f);
    const __m256 one = _mm256_set1_ps(1.0f);
    const __m256 minus_a_lot = _mm256_set1_ps(-1000.0f);
    const __m256 ro_x = _mm256_set1_ps(ray.origin.x), ro_y = _mm256_set1_ps(ray.origin.y),
                 ro_z = _mm256_set1_ps(ray.origin.z);
    const __m256 rd_x = _mm256_set1_ps(ray.direction.x), rd_y = _mm256_set1_ps(ray.direction.y),
                 rd_z = _mm256_set1_ps(ray.direction.z);

    const size_t n_chunks = m.n_total / 8;
    for (size_t c = 0; c < n_chunks; ++c) {
        const size_t o = c * 8;
        __m256 vax = _mm256_loadu_ps(m.v0x + o), vay = _mm256_loadu_ps(m.v0y + o),
               vaz = _mm256_loadu_ps(m.v0z + o);
        __m256 eax = _mm256_loadu_ps(m.e1x + o), eay = _mm256_loadu_ps(m.e1y + o),
               eaz = _mm256_loadu_ps(m.e1z + o);
        __m256 ebx = _mm256_loadu_ps(m.e2x + o), eby = _mm256_loadu_ps(m.e2y + o),
               ebz = _mm256_loadu_ps(m.e2z + o);

        __m256 hx, hy, hz;
        avx_cross(rd_x, rd_y, rd_z, ebx, eby, ebz, hx, hy, hz);       // h = d x e_b
        __m256 a = avx_dot(eax, eay, eaz, hx, hy, hz);                 // a = e_a . h
        __m256 c1 = _mm256_and_ps(_mm256_cmp_ps(neg_eps, a, _CMP_LT_OQ),  // -eps < a && a < eps
                                  _mm256_cmp_ps(a, eps, _CMP_LT_OQ));
        __m256 f = _mm256_div_ps(one, a);
        __m256 sx = _mm256_sub_ps(ro_x, vax), sy = _mm256_sub_ps(ro_y, vay),
               sz = _mm256_sub_ps(ro_z, vaz);
        __m256 u = _mm256_mul_ps(f, avx_dot(sx, sy, sz, hx, hy, hz));
        __m256 c2 = _mm256_or_ps(_mm256_cmp_ps(u, zero, _CMP_LT_OQ), _mm256_cmp_ps(u, one, _CMP_GT_OQ));
        __m256 qx, qy, qz;
        avx_cross(sx, sy, sz, eax, eay, eaz, qx, qy, qz);             // q = s x e_a
        __m256 v = _mm256_mul_ps(f, avx_dot(rd_x, rd_y, rd_z, qx, qy, qz));
        __m256 c3 = _mm256_or_ps(_mm256_cmp_ps(v, zero, _CMP_LT_OQ),
                                 _mm256_cmp_ps(_mm256_add_ps(u, v), one, _CMP_GT_OQ));
        __m256 t = _mm256_mul_ps(f, avx_dot(ebx, eby, ebz, qx, qy, qz));
        __m256 c4 = _mm256_and_ps(_mm256_cmp_ps(t, eps, _CMP_GT_OQ),
                                  _mm256_cmp_ps(t, eps_frac, _CMP_LT_OQ));
        __m256 rejected = _mm256_or_ps(c1, _mm256_or_ps(c2, c3));
        __m256 has = _mm256_andnot_ps(rejected, c4);
        __m256 res = _mm256_or_ps(_mm256_and_ps(has, t), _mm256_andnot_ps(has, minus_a_lot));
        float unpacked[8];
        _mm256_storeu_ps(unpacked, res);
        ray_params.insert(ray_params.end(), unpacked, unpacked + 8);
    }
    bool ok = find_smallest_element_bigger_than_eps(ray_params.data(), ray_params.size(),
                                                    m.is_padding, eps_f32, t_out, idx_out);
    if (keep_params) keep_params->swap(ray_params);
    return ok;
}

// mesh.rs:225-267
inline bool mesh_hit(const rbrt_mesh_t& m, const Ray& ray, float min_dist, float max_dist, Hit& out,
                     float* t_out = nullptr, size_t* idx_out = nullptr) {
    if (!bbox_hit(m.bbox_lo, m.bbox_hi, ray)) return false;
    float t;
    size_t idx;
    if (!triangle_soa_avx_intersect_with_ray(ray, m, min_dist, t, idx)) return false;
    V3 p = ray.point_at(t);
    float dist = length(ray.origin - p);
    if (dist > min_dist && dist < max_dist) {
        out.point = p;
        out.normal = v3(m.nx[idx], m.ny[idx], m.nz[idx]);
        out.mat = &m.mat;
        out.dist = dist;
        if (t_out) *t_out = t;
        if (idx_out) *idx_out = idx;
        return true;
    }
    return false;
}

struct HitIds {
    float t;
    int32_t obj, tri;
};

// triangle.rs:30-34
inline V3 get_triangle_normal(const float c[3][3]) {
    V3 edge1 = v3(c[1]) - v3(c[0]);
    V3 edge2 = v3(c[2]) - v3(c[0]);
    return normalize(cross(edge1, edge2));
}

// triangle.rs:92-130 basic_triangle_intersect_w_ray + triangle.rs:412-441 (Intersectable for BasicTriangle).
// edges = [c1 - c0, c2 - c0] (triangle.rs:25); `(0.0..=1.0).contains(&u)` is false for a NaN u.
inline bool basic_triangle_hit(const rbrt_triangle_t& tri, const Ray& ray, float min_dist, float max_dist, Hit& out, float* t_out) {
    const float eps = min_dist;
    const V3 v0 = v3(tri.corners[0]);
    const V3 e0 = v3(tri.corners[1]) - v0, e1 = v3(tri.corners[2]) - v0;
    const V3 h = cross(ray.direction, e1);
    const float a = dot(e0, h);
    if (-eps < a && a < eps) return false;
    const float f = 1.0f / a;
    const V3 s = ray.origin - v0;
    const float u = f * dot(s, h);
    if (!(0.0f <= u && u <= 1.0f)) return false;
    const V3 q = cross(s, e0);
    const float v = f * dot(ray.direction, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = f * dot(e1, q);
    if (!(t > eps)) return false;
    const V3 p = ray.point_at(t);
    const float dist = length(ray.origin - p);
    if (dist < min_dist || dist > max_dist) return false;  // (checked twice in the reference: :122 and :427)
    out.point = p;
    out.normal = get_triangle_normal(tri.corners);  // BasicTriangle::new stores it (triangle.rs:23); never flipped
    out.mat = &tri.mat;
    out.dist = dist;
    if (t_out) *t_out = t;
    return true;
}

// scene.rs:19-43 — Scene::elements in their order (spheres and BasicTriangles; rbrt_scene_t::element_order), then the
// meshes; strictly smaller distance wins, so the earlier object keeps a tie.
inline bool scene_hit(const rbrt_scene_t& sc, const Ray& ray, float min_dist, float max_dist,
                      Hit& best, HitIds* ids = nullptr) {
    bool any = false;
    float closest = std::numeric_limits<float>::max();
    const uint32_t n_elem = sc.n_spheres + sc.n_triangles;
    for (uint32_t e = 0; e < n_elem; ++e) {
        const uint32_t desc = sc.element_order ? sc.element_order[e] : (e < sc.n_spheres ? e : (0x80000000u | (e - sc.n_spheres)));
        Hit h;
        float t;
        const bool hit = (desc >> 31) ? basic_triangle_hit(sc.triangles[desc & 0x7FFFFFFFu], ray, min_dist, max_dist, h, &t)
                                      : sphere_hit(sc.spheres[desc], ray, min_dist, max_dist, h, &t);
        if (hit) {
            if (h.dist < closest) {
                closest = h.dist;
                best = h;
                any = true;
                if (ids) *ids = HitIds{t, int32_t(e), -1};
            }
        }
    }
    for (uint32_t i = 0; i < sc.n_meshes; ++i) {
        Hit h;
        float t;
        size_t idx;
        if (mesh_hit(sc.meshes[i], ray, min_dist, max_dist, h, &t, &idx)) {
            if (h.dist < closest) {
                closest = h.dist;
                best = h;
                any = true;
                if (ids) *ids = HitIds{t, int32_t(n_elem + i), int32_t(idx)};
            }
        }
    }
    return any;
}

// ---------------------------------------------------------------------------------------------
// Camera + integrator  (cam.rs:64-82, lib.rs:43-73)
// ---------------------------------------------------------------------------------------------
inline Ray get_ray_through_pixel(const rbrt_camera_t& cam, uint32_t row, uint32_t col, Rng& rng) {
    float col_off = float(col) - float(cam.img_width_pix / 2);
    float row_off = float(row) - float(cam.img_height_pix / 2);
    float u0 = rng.next_f32();
    float col_mm = (col_off + u0 - 0.5f) * cam.mm_per_pix_hor;
    float u1 = rng.next_f32();
    float row_mm = (row_off + u1 - 0.5f) * cam.mm_per_pix_vert;
    V3 target = v3(cam.img_center_point) + 0.001f * col_mm * v3(cam.right) - 0.001f * row_mm * v3(cam.up);
    V3 pos = v3(cam.position);
    return Ray{pos, normalize(target - pos)};
}

V3 colorize(const Ray& ray, const rbrt_scene_t& sc, const rbrt_render_opts_t& o, V3 bg, uint32_t depth,
            Rng& rng, uint64_t* rays) {
    Hit h;
    if (rays) ++*rays;
    if (scene_hit(sc, ray, o.min_dist, o.max_dist, h)) {
        Ray scattered{v3(0, 0, 0), v3(0, 0, 0)};
        V3 att = v3(0, 0, 0);
        if (depth > 0 && scatter(*h.mat, ray, h, att, scattered, rng)) {
            return att * colorize(scattered, sc, o, bg, depth - 1, rng, rays);
        }
        return v3(0, 0, 0);
    }
    float t = 0.5f * (ray.direction.y + 1.0f);
    return t * v3(1.0f, 1.0f, 1.0f) + (1.0f - t) * bg;
}

// lib.rs:116-122: Rust `as u8` saturates, NaN -> 0.
inline uint8_t quantise(float c) {
    float v = std::sqrt(c) * 256.0f;
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return uint8_t(v);
}

}  // namespace

// =============================================================================================
// C entry points (ctypes from tests/, bench.py cpu_baseline, __graft_entry__.smoke)
// =============================================================================================
extern "C" {

// lib.rs:75-124 restricted to columns [c0,c1) x rows [r0,r1) (full image: 0,W,0,H). Pixels outside
// the window are left untouched; col_stride > 1 renders only every col_stride-th column of the window
// (an unbiased sample of the image's workload for the CPU-baseline timing). Threads pull columns from a shared counter (rayon's par_iter over
// columns, lib.rs:84-86); per-pixel arithmetic is sequential so the result is thread-count independent.
// Returns the number of Scene::hit calls made (for the CPU-baseline report).
uint64_t rbrt_oracle_render_window(const rbrt_camera_t* cam, const rbrt_scene_t* scene,
                                   const rbrt_render_opts_t* opts, uint32_t c0, uint32_t c1,
                                   uint32_t r0, uint32_t r1, uint32_t col_stride, int n_threads,
                                   float* out_radiance, uint8_t* out_rgb8) {
    const uint32_t W = cam->img_width_pix, H = cam->img_height_pix;
    if (c1 > W) c1 = W;
    if (r1 > H) r1 = H;
    if (n_threads <= 0) n_threads = int(std::thread::hardware_concurrency());
    if (n_threads <= 0) n_threads = 1;
    if (col_stride == 0) col_stride = 1;
    std::atomic<uint32_t> next_col{0};
    std::atomic<uint64_t> total_rays{0};
    const V3 bg = v3(opts->bg);
    auto worker = [&]() {
        uint64_t rays = 0;
        for (;;) {
            uint32_t col = c0 + next_col.fetch_add(1) * col_stride;
            if (col >= c1) break;
            for (uint32_t row = r0; row < r1; ++row) {
                V3 color = v3(0, 0, 0);
                for (uint32_t s = 0; s < opts->spp; ++s) {
                    Rng rng(opts->seed, row * W + col, s);
                    Ray ray = get_ray_through_pixel(*cam, row, col, rng);
                    color = color + colorize(ray, *scene, *opts, bg, opts->max_depth, rng, &rays);
                }
                color = color * (1.0f / float(opts->spp));
                size_t o = (size_t(row) * W + col) * 3;
                if (out_radiance) {
                    out_radiance[o + 0] = color.x;
                    out_radiance[o + 1] = color.y;
                    out_radiance[o + 2] = color.z;
                }
                if (out_rgb8) {
                    out_rgb8[o + 0] = quantise(color.x);
                    out_rgb8[o + 1] = quantise(color.y);
                    out_rgb8[o + 2] = quantise(color.z);
                }
            }
        }
        total_rays.fetch_add(rays);
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    return total_rays.load();
}

uint64_t rbrt_oracle_render(const rbrt_camera_t* cam, const rbrt_scene_t* scene,
                            const rbrt_render_opts_t* opts, int n_threads, float* out_radiance,
                            uint8_t* out_rgb8) {
    return rbrt_oracle_render_window(cam, scene, opts, 0, cam->img_width_pix, 0, cam->img_height_pix, 1,
                                     n_threads, out_radiance, out_rgb8);
}

uint64_t rbrt_oracle_nan_discriminants(void) { return g_nan_discriminants.load(); }

// Scene::hit for a batch of rays (same outputs as rbrt_hip_trace_rays).
void rbrt_oracle_trace_rays(const rbrt_scene_t* scene, const float* rays, size_t n, float min_dist,
                            float max_dist, int n_threads, float* out_t, int32_t* out_obj,
                            int32_t* out_tri, float* out_dist) {
    if (n_threads <= 0) n_threads = int(std::thread::hardware_concurrency());
    if (n_threads <= 0) n_threads = 1;
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            size_t b = next.fetch_add(256);
            if (b >= n) break;
            size_t e = b + 256 < n ? b + 256 : n;
            for (size_t i = b; i < e; ++i) {
                Ray r{v3(rays + 6 * i), v3(rays + 6 * i + 3)};
                Hit h;
                HitIds ids{std::numeric_limits<float>::quiet_NaN(), -1, -1};
                bool ok = scene_hit(*scene, r, min_dist, max_dist, h, &ids);
                if (out_t) out_t[i] = ok ? ids.t : std::numeric_limits<float>::quiet_NaN();
                if (out_obj) out_obj[i] = ok ? ids.obj : -1;
                if (out_tri) out_tri[i] = ok ? ids.tri : -1;
                if (out_dist) out_dist[i] = ok ? h.dist : std::numeric_limits<float>::quiet_NaN();
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
}

// ---- scene preparation (cold path of the reference, restated so the C++ host can be checked) ----

// cam.rs:22-62
void rbrt_oracle_camera_new(const float position[3], const float look_at[3], const float up[3],
                            uint32_t img_height_pix, uint32_t img_width_pix, float focal_len_mm,
                            rbrt_camera_t* out) {
    V3 pos = v3(position), la = v3(look_at), upv = v3(up);
    V3 right = normalize(cross(normalize(la), normalize(upv)));
    float img_width_mm = 35.0f;
    float mm_per_pix_hor = img_width_mm / float(img_width_pix);
    float img_height_mm = float(img_height_pix) * mm_per_pix_hor;
    float mm_per_pix_vert = img_height_mm / float(img_height_pix);
    V3 center = pos + focal_len_mm / 1000.0f * normalize(la);
    out->position[0] = pos.x, out->position[1] = pos.y, out->position[2] = pos.z;
    out->right[0] = right.x, out->right[1] = right.y, out->right[2] = right.z;
    out->up[0] = upv.x, out->up[1] = upv.y, out->up[2] = upv.z;
    out->img_center_point[0] = center.x, out->img_center_point[1] = center.y,
    out->img_center_point[2] = center.z;
    out->mm_per_pix_hor = mm_per_pix_hor;
    out->mm_per_pix_vert = mm_per_pix_vert;
    out->img_width_pix = img_width_pix;
    out->img_height_pix = img_height_pix;
}

// mesh.rs:41-74 + 102-112 + 123-181: from raw .obj triangles (n x 9 floats: v0 v1 v2) to the SoA
// arrays. Every output array must hold n + n%8 entries. Returns n_total.
uint32_t rbrt_oracle_mesh_prep(const float* tri_vertices, uint32_t n, float scale,
                               const float rotation[3], const float translation[3], float* v0x,
                               float* v0y, float* v0z, float* e1x, float* e1y, float* e1z, float* e2x,
                               float* e2y, float* e2z, float* nx, float* ny, float* nz,
                               uint8_t* is_padding, float bbox_lo[3], float bbox_hi[3]) {
    V3 rot = v3(rotation), tr = v3(translation);
    std::vector<V3> a(n), b(n), c(n);
    for (uint32_t i = 0; i < n; ++i) {
        const float* p = tri_vertices + 9 * size_t(i);
        V3 vs[3];
        for (int k = 0; k < 3; ++k) {
            V3 scaled = v3(p[3 * k] * scale, p[3 * k + 1] * scale, p[3 * k + 2] * scale);
            vs[k] = rotate_point(scaled, rot) + tr;
        }
        a[i] = vs[0], b[i] = vs[1], c[i] = vs[2];
    }
    // aabbox.rs:62-88 (over the real triangles, before padding: mesh.rs:62)
    V3 lo = v3(std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
               std::numeric_limits<float>::max());
    V3 hi = v3(-std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(),
               -std::numeric_limits<float>::max());
    for (uint32_t i = 0; i < n; ++i) {
        const V3 vs[3] = {a[i], b[i], c[i]};
        for (const V3& v : vs) {
            if (v.x < lo.x) lo.x = v.x;
            if (v.y < lo.y) lo.y = v.y;
            if (v.z < lo.z) lo.z = v.z;
            if (v.x > hi.x) hi.x = v.x;
            if (v.y > hi.y) hi.y = v.y;
            if (v.z > hi.z) hi.z = v.z;
        }
    }
    bbox_lo[0] = lo.x, bbox_lo[1] = lo.y, bbox_lo[2] = lo.z;
    bbox_hi[0] = hi.x, bbox_hi[1] = hi.y, bbox_hi[2] = hi.z;
    const uint32_t n_pad = n % 8;  // mesh.rs:136 with 8 lanes (AVX, mesh.rs:28-30)
    for (uint32_t i = 0; i < n + n_pad; ++i) {
        uint32_t src = i < n ? i : 0;  // mesh.rs:139-143: copies of triangle 0
        V3 ea = b[src] - a[src], eb = c[src] - a[src];
        V3 nn = normalize(cross(ea, eb));  // triangle.rs:30-34
        v0x[i] = a[src].x, v0y[i] = a[src].y, v0z[i] = a[src].z;
        e1x[i] = ea.x, e1y[i] = ea.y, e1z[i] = ea.z;
        e2x[i] = eb.x, e2y[i] = eb.y, e2z[i] = eb.z;
        nx[i] = nn.x, ny[i] = nn.y, nz[i] = nn.z;
        is_padding[i] = i >= n;
    }
    return n + n_pad;
}

// ---- known-answer hooks: one per reference unit test (tests/test_oracle_kats.py) ----------------
void rbrt_oracle_kat_vec3(int op, const float a[3], const float b[3], float out[3]) {
    V3 A = v3(a), B = v3(b), r = v3(0, 0, 0);
    switch (op) {
        case 0: r = cross(A, B); break;
        case 1: r = normalize(A); break;
        case 2: r = v3(dot(A, B), 0, 0); break;
        case 3: r = A * B; break;
        case 4: r = A + B; break;
        case 5: r = A - B; break;
        case 6: r = rotate_point(A, B); break;
        case 7: r = v3(length(A), 0, 0); break;
        case 8: r = reflect(A, B); break;
        case 9: r = normalize(cross(B - A, v3(0, 0, 0))); break;  // unused
    }
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void rbrt_oracle_kat_triangle_normal(const float corners[9], float out[3]) {  // triangle.rs:30-34
    V3 c0 = v3(corners), c1 = v3(corners + 3), c2 = v3(corners + 6);
    V3 n = normalize(cross(c1 - c0, c2 - c0));
    out[0] = n.x, out[1] = n.y, out[2] = n.z;
}
int rbrt_oracle_kat_refract(const float dir[3], const float n[3], float ni_over_nt, float out[3]) {
    V3 r = v3(0, 0, 0);
    bool ok = refract(v3(dir), v3(n), ni_over_nt, r);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
    return ok;
}
float rbrt_oracle_kat_schlick(float cosine, float ref_idx) { return schlick(cosine, ref_idx); }
int rbrt_oracle_kat_sphere(const rbrt_sphere_t* s, const float ray[6], float min_dist, float max_dist,
                           float point[3], float normal[3], float* dist) {
    Hit h;
    Ray r{v3(ray), v3(ray + 3)};
    if (!sphere_hit(*s, r, min_dist, max_dist, h)) return 0;
    point[0] = h.point.x, point[1] = h.point.y, point[2] = h.point.z;
    normal[0] = h.normal.x, normal[1] = h.normal.y, normal[2] = h.normal.z;
    *dist = h.dist;
    return 1;
}
void rbrt_oracle_kat_unit_sphere(uint64_t seed, uint32_t pixel, uint32_t sample, int n, float* out) {
    Rng rng(seed, pixel, sample);
    for (int i = 0; i < n; ++i) {
        V3 p = random_point_in_unit_sphere(rng);
        out[3 * i] = p.x, out[3 * i + 1] = p.y, out[3 * i + 2] = p.z;
    }
}
void rbrt_oracle_kat_rng(uint64_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* raw,
                         float* f32) {
    Rng rng(seed, pixel, sample);
    Rng rng2(seed, pixel, sample);
    for (int i = 0; i < n; ++i) {
        if (raw) raw[i] = rng.next_u32();
        if (f32) f32[i] = rng2.next_f32();
    }
}
void rbrt_oracle_kat_avx(int op, const float* a, const float* b, float* out) {  // 3x8 in, 3x8 / 8 out
    __m256 ax = _mm256_loadu_ps(a), ay = _mm256_loadu_ps(a + 8), az = _mm256_loadu_ps(a + 16);
    __m256 bx = _mm256_loadu_ps(b), by = _mm256_loadu_ps(b + 8), bz = _mm256_loadu_ps(b + 16);
    if (op == 0) {
        __m256 cx, cy, cz;
        avx_cross(ax, ay, az, bx, by, bz, cx, cy, cz);
        _mm256_storeu_ps(out, cx), _mm256_storeu_ps(out + 8, cy), _mm256_storeu_ps(out + 16, cz);
    } else {
        _mm256_storeu_ps(out, avx_dot(ax, ay, az, bx, by, bz));
    }
}
int rbrt_oracle_kat_bbox_hit(const float lo[3], const float hi[3], const float ray[6]) {
    Ray r{v3(ray), v3(ray + 3)};
    return bbox_hit(lo, hi, r);
}
// Brute-force mesh kernel alone (T4+T5): returns 1 on hit; per-triangle t's into params (n_total/8*8).
int rbrt_oracle_kat_mesh_intersect(const rbrt_mesh_t* m, const float ray[6], float min_dist, float* t,
                                   int32_t* idx, float* params) {
    Ray r{v3(ray), v3(ray + 3)};
    std::vector<float> keep;
    float tt = 0;
    size_t ii = 0;
    bool ok = triangle_soa_avx_intersect_with_ray(r, *m, min_dist, tt, ii, &keep);
    if (params) std::memcpy(params, keep.data(), keep.size() * sizeof(float));
    *t = tt;
    *idx = int32_t(ii);
    return ok;
}
void rbrt_oracle_kat_camera_ray(const rbrt_camera_t* cam, uint32_t row, uint32_t col, uint64_t seed,
                                uint32_t sample, float out[6]) {
    Rng rng(seed, row * cam->img_width_pix + col, sample);
    Ray r = get_ray_through_pixel(*cam, row, col, rng);
    out[0] = r.origin.x, out[1] = r.origin.y, out[2] = r.origin.z;
    out[3] = r.direction.x, out[4] = r.direction.y, out[5] = r.direction.z;
}
// One scatter event with an explicit stream: returns the bool of RayScattering::scatter.
int rbrt_oracle_kat_scatter(const rbrt_material_t* m, const float ray[6], const float point[3],
                            const float normal[3], uint64_t seed, uint32_t pixel, uint32_t sample,
                            float att[3], float out_ray[6]) {
    Rng rng(seed, pixel, sample);
    Ray in{v3(ray), v3(ray + 3)};
    Hit h{v3(point), v3(normal), m, 0.0f};
    V3 a = v3(0, 0, 0);
    Ray o{v3(0, 0, 0), v3(0, 0, 0)};
    bool ok = scatter(*m, in, h, a, o, rng);
    att[0] = a.x, att[1] = a.y, att[2] = a.z;
    out_ray[0] = o.origin.x, out_ray[1] = o.origin.y, out_ray[2] = o.origin.z;
    out_ray[3] = o.direction.x, out_ray[4] = o.direction.y, out_ray[5] = o.direction.z;
    return ok;
}
// The same event, also reporting the stream's state before and after it ({s0, s1} each): what the HIP hook
// rbrt_hip_debug_scatter takes and returns, so that a fixture can check the draw COUNT of an event as well.
int rbrt_oracle_kat_scatter_state(const rbrt_material_t* m, const float ray[6], const float point[3],
                                  const float normal[3], uint64_t seed, uint32_t pixel, uint32_t sample,
                                  float att[3], float out_ray[6], uint32_t state_before[2], uint32_t state_after[2]) {
    Rng rng(seed, pixel, sample);
    state_before[0] = rng.s0, state_before[1] = rng.s1;
    Ray in{v3(ray), v3(ray + 3)};
    Hit h{v3(point), v3(normal), m, 0.0f};
    V3 a = v3(0, 0, 0);
    Ray o{v3(0, 0, 0), v3(0, 0, 0)};
    bool ok = scatter(*m, in, h, a, o, rng);
    att[0] = a.x, att[1] = a.y, att[2] = a.z;
    out_ray[0] = o.origin.x, out_ray[1] = o.origin.y, out_ray[2] = o.origin.z;
    out_ray[3] = o.direction.x, out_ray[4] = o.direction.y, out_ray[5] = o.direction.z;
    state_after[0] = rng.s0, state_after[1] = rng.s1;
    return ok;
}
// BasicTriangle::intersect_with_ray alone (triangle.rs:412-441): returns hit, t, distance, normal.
int rbrt_oracle_kat_basic_triangle(const rbrt_triangle_t* tri, const float ray[6], float min_dist, float max_dist, float* t,
                                   float* dist, float normal[3]) {
    Hit h;
    float tt = 0.0f;
    Ray r{v3(ray), v3(ray + 3)};
    if (!basic_triangle_hit(*tri, r, min_dist, max_dist, h, &tt)) return 0;
    *t = tt, *dist = h.dist;
    normal[0] = h.normal.x, normal[1] = h.normal.y, normal[2] = h.normal.z;
    return 1;
}
uint8_t rbrt_oracle_kat_quantise(float c) { return quantise(c); }

}  // extern "C"
