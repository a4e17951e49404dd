// bvh_device.hip — BVH construction on the GPU (gfx950): Morton-ordered binary radix tree (Karras 2012), bottom-up
// box fit, collapse into the same 4-wide nodes and <= kLeafMax-triangle leaves the host builder (bvh.cpp) emits.
//
// Why: the host builder's binned-SAH tree takes 0.4 s for an 871k-triangle mesh, two orders of magnitude more than
// rendering a frame of it. The reference has no acceleration structure at all (mesh.rs:232-243), so
// "any tree is correct" as long as bvh.h's contract holds: exactly the triangles the scan can return are indexed
// (i < 8*floor(n_total/8), !is_padding[i]: triangle.rs:166-167, :400), every child box contains its triangles, every
// child carries max |e1|*|e2| of its subtree (the traversal's culling pad). The parity tests (GPU image == brute-force
// oracle, trace_rays == scan) are the acceptance test for whichever builder made the tree.
//
// Stages (all on one stream, no host round trip until the final 32-byte read-back):
//   prim_setup   per SoA entry: validity, box, centroid, |e1||e2|; centroid bounds by atomic min/max; valid count
//   morton_keys  63-bit Morton code of the centroid (21 bits per axis) | invalid entries sort last
//   rocprim::radix_sort_pairs (key = code, value = reference index)
//   radix_tree   one thread per internal node of the binary radix tree (ties broken by position)
//   fit          bottom-up: second arrival at a node merges its children's boxes
//   collapse     level by level from the root: a binary subtree of <= kLeafMax triangles becomes a leaf, larger
//                ones are opened (largest surface first) until a node has four children; unused slots get NaN boxes
//   emit_tris    48-byte records in leaf order, links absolute in the scene's triangle array
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>  // (rocprim 7.2 headers use memset without including it)

#include <rocprim/device/device_radix_sort.hpp>

#include "bvh_device.h"
#include "device_types.h"

namespace rbrt {
namespace {

constexpr int kTpb = 256;

struct Bin {  // internal node i of the binary radix tree (n_valid - 1 of them); leaves are positions in sorted order
    int32_t left, right;  // >= 0: internal node; < 0: ~position of a leaf
    int32_t parent;
    uint32_t first, last;  // covered positions [first, last]
};

struct Work {  // build state shared by the kernels
    uint32_t n_valid;
    uint32_t cmin[3], cmax[3];  // centroid bounds, order-preserving uint encoding of the floats
    uint32_t n_nodes;           // 4-wide nodes allocated so far
    uint32_t overflow;          // bit 0: node array full; bit 1: depth budget exceeded
    uint32_t max_depth;
    uint32_t q_count[2];        // collapse queues
    float max_e12;
};

__device__ __forceinline__ uint32_t fenc(float f) {  // order-preserving float -> uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fdec(uint32_t e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}

struct Soa {
    const float *v0x, *v0y, *v0z, *e1x, *e1y, *e1z, *e2x, *e2y, *e2z;
    const uint8_t* is_padding;
};

// boxes: [n_total][8] = lo.xyz, e12, hi.xyz, valid
__global__ __launch_bounds__(kTpb) void prim_setup(Soa m, uint32_t n_total, uint32_t n_tested, float* __restrict__ boxes, Work* w) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    bool valid = i < n_tested;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, c[3] = {0, 0, 0}, e12 = 0.0f;
    if (valid) {
        const float v0[3] = {m.v0x[i], m.v0y[i], m.v0z[i]};
        const float e1[3] = {m.e1x[i], m.e1y[i], m.e1z[i]};
        const float e2[3] = {m.e2x[i], m.e2y[i], m.e2z[i]};
        valid = m.is_padding[i] == 0;
        const float fmax = 3.40282347e+38f;
        for (int k = 0; k < 3; ++k) {
            // non-finite inputs can never pass the ordered compares of triangle.rs:198-241: not indexed (as bvh.cpp)
            valid = valid && __builtin_isfinite(v0[k]) && __builtin_isfinite(e1[k]) && __builtin_isfinite(e2[k]);
            const float v1 = __builtin_fminf(__builtin_fmaxf(v0[k] + e1[k], -fmax), fmax);
            const float v2 = __builtin_fminf(__builtin_fmaxf(v0[k] + e2[k], -fmax), fmax);
            lo[k] = __builtin_fminf(v0[k], __builtin_fminf(v1, v2));
            hi[k] = __builtin_fmaxf(v0[k], __builtin_fmaxf(v1, v2));
            c[k] = 0.5f * lo[k] + 0.5f * hi[k];
        }
        const float l1 = __builtin_sqrtf(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
        const float l2 = __builtin_sqrtf(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        e12 = l1 * l2;
    }
    if (i < n_total) {
        float4* b = reinterpret_cast<float4*>(boxes + size_t(i) * 8u);
        b[0] = make_float4(lo[0], lo[1], lo[2], e12);
        b[1] = make_float4(hi[0], hi[1], hi[2], valid ? 1.0f : 0.0f);
    }
    // centroid bounds + count: wave reduction, one atomic per wave
    const uint64_t vm = __builtin_amdgcn_ballot_w64(valid);
    if (vm == 0ull) return;
    for (int k = 0; k < 3; ++k) {
        uint32_t mn = valid ? fenc(c[k]) : 0xFFFFFFFFu, mx = valid ? fenc(c[k]) : 0u;
        for (int off = 32; off > 0; off >>= 1) {
            mn = min(mn, uint32_t(__shfl_xor(int(mn), off)));
            mx = max(mx, uint32_t(__shfl_xor(int(mx), off)));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&w->cmin[k], mn);
            atomicMax(&w->cmax[k], mx);
        }
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&w->n_valid, uint32_t(__popcll(vm)));
}

__device__ __forceinline__ uint64_t spread21(uint32_t x) {  // 21 bits -> every third bit of 63
    uint64_t v = x & 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(kTpb) void morton_keys(const float* __restrict__ boxes, uint32_t n_total, const Work* w,
                                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    if (i >= n_total) return;
    const float4 a = reinterpret_cast<const float4*>(boxes + size_t(i) * 8u)[0];
    const float4 b = reinterpret_cast<const float4*>(boxes + size_t(i) * 8u)[1];
    uint64_t key = ~0ull;  // invalid entries sort behind every valid one
    if (b.w != 0.0f) {
        const float lo[3] = {a.x, a.y, a.z}, hi[3] = {b.x, b.y, b.z};
        uint32_t q[3];
        for (int k = 0; k < 3; ++k) {
            const float cmin = fdec(w->cmin[k]), cmax = fdec(w->cmax[k]);
            const float c = 0.5f * lo[k] + 0.5f * hi[k];
            const float ext = cmax - cmin;
            float t = ext > 0.0f ? (c - cmin) / ext : 0.0f;  // (inf extent: t = 0 or NaN -> clamped below)
            t = __builtin_fminf(__builtin_fmaxf(t, 0.0f), 1.0f);
            q[k] = uint32_t(t * 2097151.0f);
        }
        key = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);  // < 2^63
    }
    keys[i] = key;
    vals[i] = i;
}

// length of the common prefix of the keys at positions i and j (ties: the positions themselves), -1 outside the range
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz(uint32_t(i) ^ uint32_t(j));
    return __clzll(a ^ b);
}

__global__ __launch_bounds__(kTpb) void radix_tree(const uint64_t* __restrict__ keys, const Work* w, Bin* __restrict__ bin,
                                                   int32_t* __restrict__ leaf_parent) {
    const int n = int(w->n_valid);
    const int i = blockIdx.x * kTpb + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int32_t left = first == gamma ? ~gamma : gamma;
    const int32_t right = last == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    // (field by field: a node's `parent` is written by its parent's thread, the rest by its own)
    bin[i].left = left, bin[i].right = right, bin[i].first = uint32_t(first), bin[i].last = uint32_t(last);
    if (left >= 0) bin[left].parent = i; else leaf_parent[~left] = i;
    if (right >= 0) bin[right].parent = i; else leaf_parent[~right] = i;
    if (i == 0) bin[0].parent = -1;
}

// nbox: [n_valid - 1][8] = lo.xyz, max e12, hi.xyz, half area
__global__ __launch_bounds__(kTpb) void fit(const Work* w, const Bin* __restrict__ bin, const int32_t* __restrict__ leaf_parent,
                                            const uint32_t* __restrict__ sorted, const float* __restrict__ boxes,
                                            float* __restrict__ nbox, uint32_t* __restrict__ arrived) {
    const uint32_t n = w->n_valid;
    const uint32_t p = blockIdx.x * kTpb + threadIdx.x;
    if (p >= n || n < 2) return;
    int32_t node = leaf_parent[p];
    while (node >= 0) {
        __threadfence();  // the first arrival's box writes (made by another thread) before this thread reads them
        if (atomicAdd(&arrived[node], 1u) == 0u) return;  // the second arrival does the merge
        __threadfence();
        float lo[3], hi[3], e = 0.0f;
        for (int k = 0; k < 3; ++k) lo[k] = 3.40282347e+38f, hi[k] = -3.40282347e+38f;
        const int32_t ch[2] = {bin[node].left, bin[node].right};
        for (int c = 0; c < 2; ++c) {
            const float* src = ch[c] >= 0 ? nbox + size_t(ch[c]) * 8u : boxes + size_t(sorted[~ch[c]]) * 8u;
            float v[8];  // (a node's box was written by a thread of possibly another workgroup: after the fences above)
            for (int k = 0; k < 8; ++k) v[k] = src[k];
            for (int k = 0; k < 3; ++k) lo[k] = __builtin_fminf(lo[k], v[k]), hi[k] = __builtin_fmaxf(hi[k], v[4 + k]);
            e = __builtin_fmaxf(e, v[3]);
        }
        float* dst = nbox + size_t(node) * 8u;
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        dst[0] = lo[0], dst[1] = lo[1], dst[2] = lo[2], dst[3] = e;
        dst[4] = hi[0], dst[5] = hi[1], dst[6] = hi[2], dst[7] = dx * dy + dy * dz + dz * dx;
        node = bin[node].parent;
    }
}

struct QItem {
    int32_t bnode;   // binary node to turn into a 4-wide node
    uint32_t out;    // its index in the 4-wide node array
};

__device__ __forceinline__ void cand_box(int32_t ref, const float* __restrict__ nbox, const float* __restrict__ boxes,
                                         const uint32_t* __restrict__ sorted, float v[8]) {
    const float* src = ref >= 0 ? nbox + size_t(ref) * 8u : boxes + size_t(sorted[~ref]) * 8u;
    for (int k = 0; k < 8; ++k) v[k] = src[k];
    if (ref < 0) {  // a single triangle: half area from its box
        const float dx = v[4] - v[0], dy = v[5] - v[1], dz = v[6] - v[2];
        v[7] = dx * dy + dy * dz + dz * dx;
    }
}

// One level of the collapse. A candidate is a binary node (>= 0) or a single leaf position (~p < 0). A binary node
// covering <= kLeafMax positions becomes a leaf; a larger one may be opened into its two children.
__global__ __launch_bounds__(kTpb) void collapse_level(Work* w, const Bin* __restrict__ bin, const float* __restrict__ nbox,
                                                       const float* __restrict__ boxes, uint32_t* __restrict__ sorted,
                                                       const QItem* __restrict__ q_in, uint32_t n_in, QItem* __restrict__ q_out,
                                                       uint32_t* q_out_count, BvhNode4* __restrict__ nodes, uint32_t node_cap,
                                                       uint32_t tri_base, uint32_t depth) {
    const uint32_t t = blockIdx.x * kTpb + threadIdx.x;
    if (t >= n_in) return;
    const QItem it = q_in[t];
    int32_t c[4] = {bin[it.bnode].left, bin[it.bnode].right, 0, 0};
    int n = 2;
    auto expandable = [&](int32_t ref) { return ref >= 0 && bin[ref].last - bin[ref].first + 1u > uint32_t(kLeafMax); };
    while (n < 4) {  // open the expandable candidate with the largest surface (as bvh.cpp's collapse does)
        int pick = -1;
        float best = -1.0f;
        for (int k = 0; k < n; ++k)
            if (expandable(c[k])) {
                const float a = nbox[size_t(c[k]) * 8u + 7u];
                if (a > best) best = a, pick = k;
            }
        if (pick < 0) break;
        const int32_t open = c[pick];
        c[pick] = bin[open].left;
        c[n++] = bin[open].right;
    }
    BvhNode4 o;
    const float qnan = __uint_as_float(0x7fc00000u);
    for (int k = 0; k < 4; ++k) {
        if (k >= n) {
            o.lo_x[k] = o.lo_y[k] = o.lo_z[k] = o.hi_x[k] = o.hi_y[k] = o.hi_z[k] = qnan;
            o.child[k] = kNoChild;
            o.max_e12[k] = 0.0f;
            continue;
        }
        float v[8];
        cand_box(c[k], nbox, boxes, sorted, v);
        o.lo_x[k] = v[0], o.lo_y[k] = v[1], o.lo_z[k] = v[2];
        o.hi_x[k] = v[4], o.hi_y[k] = v[5], o.hi_z[k] = v[6];
        o.max_e12[k] = v[3];
        if (expandable(c[k])) {  // an inner child: allocate its 4-wide node, queue it for the next level
            const uint32_t idx = atomicAdd(&w->n_nodes, 1u);
            if (idx >= node_cap || depth + 1u > uint32_t(kMaxBvhDepth)) {
                atomicOr(&w->overflow, idx >= node_cap ? 1u : 2u);
                o.child[k] = kNoChild;  // (the build is discarded: the host builder takes over)
                continue;
            }
            o.child[k] = int32_t(idx);
            q_out[atomicAdd(q_out_count, 1u)] = QItem{c[k], idx};
            atomicMax(&w->max_depth, depth + 1u);
        } else {  // a leaf: positions [first, last] of the sorted order = records tri_base + first .. in the scene's array
            const uint32_t first = c[k] >= 0 ? bin[c[k]].first : uint32_t(~c[k]);
            const uint32_t last = c[k] >= 0 ? bin[c[k]].last : uint32_t(~c[k]);
            for (uint32_t a = first + 1; a <= last; ++a) {  // ascending reference index inside a leaf (as bvh.cpp)
                const uint32_t x = sorted[a];
                uint32_t b = a;
                while (b > first && sorted[b - 1] > x) sorted[b] = sorted[b - 1], --b;
                sorted[b] = x;
            }
            o.child[k] = ~int32_t(((tri_base + first) << kLeafBits) | (last - first));
        }
    }
    nodes[it.out] = o;
}

__global__ __launch_bounds__(kTpb) void emit_tris(Soa m, const Work* w, const uint32_t* __restrict__ sorted, BvhTri* __restrict__ out) {
    const uint32_t p = blockIdx.x * kTpb + threadIdx.x;
    if (p >= w->n_valid) return;
    const uint32_t i = sorted[p];
    BvhTri t;
    t.v0[0] = m.v0x[i], t.v0[1] = m.v0y[i], t.v0[2] = m.v0z[i];
    t.e1x = m.e1x[i], t.e1yz[0] = m.e1y[i], t.e1yz[1] = m.e1z[i];
    t.e2xy[0] = m.e2x[i], t.e2xy[1] = m.e2y[i], t.e2z = m.e2z[i];
    t.index = i;
    t.pad[0] = t.pad[1] = 0;
    out[p] = t;
}

__global__ __launch_bounds__(kTpb) void normals4(const float* __restrict__ nx, const float* __restrict__ ny, const float* __restrict__ nz,
                                                 uint32_t n, Normal4* __restrict__ out) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    if (i < n) out[i] = Normal4{nx[i], ny[i], nz[i], 0.0f};
}

__global__ void init_work(Work* w) {
    w->n_valid = 0;
    for (int k = 0; k < 3; ++k) w->cmin[k] = 0xFFFFFFFFu, w->cmax[k] = 0u;
    w->n_nodes = 1;  // the root
    w->overflow = 0;
    w->max_depth = 0;
    w->q_count[0] = w->q_count[1] = 0;
    w->max_e12 = 0.0f;
}
__global__ void root_info(Work* w, const float* nbox) { w->max_e12 = nbox[3]; }

#define DEV_TRY(expr)                 \
    do {                              \
        hipError_t _e = (expr);       \
        if (_e != hipSuccess) {       \
            cleanup();                \
            return _e;                \
        }                             \
    } while (0)

inline uint32_t blocks(uint32_t n) { return (n + kTpb - 1) / kTpb; }

}  // namespace

hipError_t device_normals(const float* d_nx, const float* d_ny, const float* d_nz, uint32_t n, Normal4* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(normals4, dim3(blocks(n)), dim3(kTpb), 0, stream, d_nx, d_ny, d_nz, n, d_out);
    return hipGetLastError();
}

// See bvh_device.h.
hipError_t build_bvh_device(const DeviceMeshSoa& soa, uint32_t n_total, BvhTri* d_tris_out, uint32_t tri_base,
                            DeviceBvhResult* res, hipStream_t stream) {
    res->ok = false;
    res->d_nodes = nullptr;
    const uint32_t n_tested = (n_total / 8u) * 8u;  // triangle.rs:166-167
    if (n_tested < 8u) return hipSuccess;           // (callers route tiny meshes to the host builder)
    const Soa m = {soa.v0x, soa.v0y, soa.v0z, soa.e1x, soa.e1y, soa.e1z, soa.e2x, soa.e2y, soa.e2z, soa.is_padding};
    const uint32_t node_cap = n_tested / 2u + 8u;
    Work* w = nullptr;
    float *boxes = nullptr, *nbox = nullptr;
    uint64_t *keys = nullptr, *keys2 = nullptr;
    uint32_t *vals = nullptr, *sorted = nullptr, *arrived = nullptr;
    int32_t* leaf_parent = nullptr;
    Bin* bin = nullptr;
    QItem* queue[2] = {nullptr, nullptr};
    void* tmp = nullptr;
    BvhNode4* nodes = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(w), (void)hipFree(boxes), (void)hipFree(nbox), (void)hipFree(keys), (void)hipFree(keys2);
        (void)hipFree(vals), (void)hipFree(sorted), (void)hipFree(arrived), (void)hipFree(leaf_parent), (void)hipFree(bin);
        (void)hipFree(queue[0]), (void)hipFree(queue[1]), (void)hipFree(tmp);
        if (!res->ok) (void)hipFree(nodes);
    };
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&w), sizeof(Work)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&boxes), size_t(n_total) * 8u * sizeof(float)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&keys), size_t(n_total) * sizeof(uint64_t)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&keys2), size_t(n_total) * sizeof(uint64_t)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&vals), size_t(n_total) * sizeof(uint32_t)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&sorted), size_t(n_total) * sizeof(uint32_t)));
    hipLaunchKernelGGL(init_work, dim3(1), dim3(1), 0, stream, w);
    hipLaunchKernelGGL(prim_setup, dim3(blocks(n_total)), dim3(kTpb), 0, stream, m, n_total, n_tested, boxes, w);
    hipLaunchKernelGGL(morton_keys, dim3(blocks(n_total)), dim3(kTpb), 0, stream, boxes, n_total, w, keys, vals);
    size_t tmp_bytes = 0;
    DEV_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, sorted, n_total, 0, 64, stream));
    DEV_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    DEV_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, sorted, n_total, 0, 64, stream));
    // n_valid decides the sizes of everything below
    Work hw;
    DEV_TRY(hipMemcpyAsync(&hw, w, sizeof(hw), hipMemcpyDeviceToHost, stream));
    DEV_TRY(hipStreamSynchronize(stream));
    const uint32_t n = hw.n_valid;
    res->n_valid = n;
    if (n <= uint32_t(kLeafMax)) {  // nothing or a single leaf: the host builder's special cases cover it
        cleanup();
        return hipSuccess;
    }
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&bin), size_t(n) * sizeof(Bin)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&leaf_parent), size_t(n) * sizeof(int32_t)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&nbox), size_t(n) * 8u * sizeof(float)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&arrived), size_t(n) * sizeof(uint32_t)));
    DEV_TRY(hipMemsetAsync(arrived, 0, size_t(n) * sizeof(uint32_t), stream));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&queue[0]), size_t(node_cap) * sizeof(QItem)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&queue[1]), size_t(node_cap) * sizeof(QItem)));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&nodes), size_t(node_cap) * sizeof(BvhNode4)));
    hipLaunchKernelGGL(radix_tree, dim3(blocks(n - 1)), dim3(kTpb), 0, stream, keys2, w, bin, leaf_parent);
    hipLaunchKernelGGL(fit, dim3(blocks(n)), dim3(kTpb), 0, stream, w, bin, leaf_parent, sorted, boxes, nbox, arrived);
    hipLaunchKernelGGL(root_info, dim3(1), dim3(1), 0, stream, w, nbox);
    // collapse, level by level: the queue sizes come back to the host once per level (a 4-byte read each)
    const QItem root = {0, 0u};
    DEV_TRY(hipMemcpyAsync(queue[0], &root, sizeof(root), hipMemcpyHostToDevice, stream));
    uint32_t n_in = 1, depth = 0;
    uint32_t* d_q_count = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(w) + offsetof(Work, q_count));
    while (n_in != 0 && depth <= uint32_t(kMaxBvhDepth)) {
        const int cur = int(depth & 1u);
        DEV_TRY(hipMemsetAsync(d_q_count + (cur ^ 1), 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(collapse_level, dim3(blocks(n_in)), dim3(kTpb), 0, stream, w, bin, nbox, boxes, sorted, queue[cur], n_in,
                           queue[cur ^ 1], d_q_count + (cur ^ 1), nodes, node_cap, tri_base, depth);
        DEV_TRY(hipMemcpyAsync(&n_in, d_q_count + (cur ^ 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        DEV_TRY(hipStreamSynchronize(stream));
        if (n_in > node_cap) n_in = node_cap;  // (overflow is flagged in w)
        ++depth;
    }
    hipLaunchKernelGGL(emit_tris, dim3(blocks(n)), dim3(kTpb), 0, stream, m, w, sorted, d_tris_out);
    DEV_TRY(hipMemcpyAsync(&hw, w, sizeof(hw), hipMemcpyDeviceToHost, stream));
    DEV_TRY(hipStreamSynchronize(stream));
    DEV_TRY(hipGetLastError());
    if (hw.overflow != 0 || n_in != 0) {  // node array or depth budget exceeded: the caller falls back to the host builder
        cleanup();
        return hipSuccess;
    }
    res->ok = true;
    res->d_nodes = nodes;
    res->n_nodes = hw.n_nodes;
    res->max_depth = hw.max_depth;
    res->max_e12 = hw.max_e12;
    cleanup();
    return hipSuccess;
}

}  // namespace rbrt
