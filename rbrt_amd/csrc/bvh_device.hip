// bvh_device.hip — BVH construction on the GPU (gfx950): Morton order, a binary tree by parallel locally-ordered
// clustering (or a binary radix tree), collapse into the same 4-wide nodes and <= kLeafMax-triangle leaves the host
// builder (bvh.cpp) emits.
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
//   binary tree  (A, default) PLOC: rounds of nearest-neighbour search in a window of the Morton order, mutual pairs
//                merge, order-preserving compaction; (B, fallback / RBRT_BVH_DEVICE_ALGO=lbvh) binary radix tree
//                (Karras 2012, ties broken by position) + bottom-up fit (second arrival merges its children's boxes)
//   collapse     level by level from the root: a binary subtree of <= kLeafMax triangles becomes a leaf, larger
//                ones are opened (largest surface first) until a node has four children; unused slots get the empty box (lo = +inf, hi = -inf)
//   emit_tris    48-byte records in leaf order, links absolute in the scene's triangle array
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>  // (rocprim 7.2 headers use memset without including it)

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "bvh_device.h"
#include "device_types.h"

namespace rbrt {
namespace {

constexpr int kTpb = 256;

// The binary tree both constructions produce, as flat arrays over node ids: leaves are ids 0 .. n-1 (positions in
// Morton order), internal nodes ids n .. 2n-2. nbox[id] = lo.xyz, max|e1||e2|, hi.xyz, half surface area.
struct Tree {
    float* nbox;       // [2n-1][8]
    int32_t* left;     // [2n-1] (internal ids only)
    int32_t* right;
    uint32_t* size;    // [2n-1] triangles below
    int32_t* parent;   // [2n-1] (radix tree only: the bottom-up fit walks it)
};

struct Work {  // build state shared by the kernels
    uint32_t n_valid;
    uint32_t cmin[3], cmax[3];  // centroid bounds, order-preserving uint encoding of the floats
    uint32_t n_nodes;           // 4-wide nodes allocated so far
    uint32_t overflow;          // bit 0: node array full; bit 1: depth budget exceeded
    uint32_t max_depth;
    uint32_t q_count[3];        // collapse queues: level d reads [d % 3], fills [(d + 1) % 3], zeroes [(d + 2) % 3]
    uint32_t ploc_m[2];         // PLOC: clusters before / after the round (in turn)
    float max_e12;
    uint32_t n_internal;        // PLOC: internal nodes created so far
    uint32_t root;              // id of the root
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

__global__ __launch_bounds__(kTpb) void init_leaves(const Work* w, const uint32_t* __restrict__ sorted, const float* __restrict__ boxes,
                                                   Tree t) {
    const uint32_t p = blockIdx.x * kTpb + threadIdx.x;
    if (p >= w->n_valid) return;
    const float* src = boxes + size_t(sorted[p]) * 8u;
    float* dst = t.nbox + size_t(p) * 8u;
    const float dx = src[4] - src[0], dy = src[5] - src[1], dz = src[6] - src[2];
    dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
    dst[4] = src[4], dst[5] = src[5], dst[6] = src[6], dst[7] = dx * dy + dy * dz + dz * dx;
    t.size[p] = 1u;
}

// ---- construction A: binary radix tree over the Morton codes (Karras 2012) + bottom-up fit -------------------------
__global__ __launch_bounds__(kTpb) void radix_tree(const uint64_t* __restrict__ keys, const Work* w, Tree t) {
    const int n = int(w->n_valid);
    const int i = blockIdx.x * kTpb + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int s = lmax >> 1; s >= 1; s >>= 1)
        if (delta(keys, n, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int sp = 0;
    for (int s = (l + 1) >> 1;; s = (s + 1) >> 1) {
        if (delta(keys, n, i, i + (sp + s) * d) > dnode) sp += s;
        if (s == 1) break;
    }
    const int gamma = i + sp * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int32_t me = n + i;
    const int32_t left = first == gamma ? gamma : n + gamma;          // a leaf position, or internal node gamma
    const int32_t right = last == gamma + 1 ? gamma + 1 : n + gamma + 1;
    t.left[me] = left, t.right[me] = right, t.size[me] = uint32_t(last - first + 1);
    t.parent[left] = me, t.parent[right] = me;
    if (i == 0) t.parent[me] = -1;
}

__global__ __launch_bounds__(kTpb) void fit(const Work* w, Tree t, uint32_t* __restrict__ arrived) {
    const uint32_t n = w->n_valid;
    const uint32_t p = blockIdx.x * kTpb + threadIdx.x;
    if (p >= n || n < 2) return;
    int32_t node = t.parent[p];
    while (node >= 0) {
        __threadfence();  // this thread's box writes, before it announces its arrival
        if (atomicAdd(&arrived[node], 1u) == 0u) return;  // the second arrival does the merge
        __threadfence();  // ... and the other arrival's writes, before they are read
        float lo[3], hi[3], e = 0.0f;
        for (int k = 0; k < 3; ++k) lo[k] = 3.40282347e+38f, hi[k] = -3.40282347e+38f;
        const int32_t ch[2] = {t.left[node], t.right[node]};
        for (int c = 0; c < 2; ++c) {
            const float* src = t.nbox + size_t(ch[c]) * 8u;
            float v[8];
            for (int k = 0; k < 8; ++k) v[k] = src[k];
            for (int k = 0; k < 3; ++k) lo[k] = __builtin_fminf(lo[k], v[k]), hi[k] = __builtin_fmaxf(hi[k], v[4 + k]);
            e = __builtin_fmaxf(e, v[3]);
        }
        float* dst = t.nbox + size_t(node) * 8u;
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        dst[0] = lo[0], dst[1] = lo[1], dst[2] = lo[2], dst[3] = e;
        dst[4] = hi[0], dst[5] = hi[1], dst[6] = hi[2], dst[7] = dx * dy + dy * dz + dz * dx;
        node = t.parent[node];
    }
}

// ---- construction B: parallel locally-ordered clustering (PLOC, Meister & Bittner 2018) ----------------------------
// The clusters -- at first the triangles in Morton order -- are merged bottom-up: every cluster looks for the
// neighbour within kPlocRadius positions whose union with it has the smallest surface, mutual choices merge, the
// array is compacted (order kept) and the round repeats until one cluster is left. Unlike the radix tree, whose
// splits are the Morton grid's, this looks at surfaces, and gives trees close to a top-down SAH build's.
constexpr int kPlocRadius = 16;  // default search radius (RBRT_PLOC_RADIUS overrides: experiments)

// The rounds of one batch are launched without the host in between: a round's cluster count is read from device memory
// (`m_ptr`), the launches are sized by the count the host last saw (an upper bound: counts only fall).
__global__ __launch_bounds__(kTpb) void ploc_nearest(const uint32_t* __restrict__ cl, const uint32_t* __restrict__ m_ptr,
                                                    const float* __restrict__ nbox, uint32_t* __restrict__ nn, uint32_t radius) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    const uint32_t m = *m_ptr;
    if (i >= m) return;
    if (m < 2u) {  // (a round after the last merge: nothing to pair with)
        nn[i] = i;
        return;
    }
    const float4 a0 = reinterpret_cast<const float4*>(nbox + size_t(cl[i]) * 8u)[0];
    const float4 a1 = reinterpret_cast<const float4*>(nbox + size_t(cl[i]) * 8u)[1];
    const uint32_t j0 = i > radius ? i - radius : 0u;
    const uint32_t j1 = i + radius < m - 1u ? i + radius : m - 1u;
    float best = 3.40282347e+38f;
    uint32_t bj = i == 0u ? 1u : i - 1u;
    for (uint32_t j = j0; j <= j1; ++j) {
        if (j == i) continue;
        const float4 b0 = reinterpret_cast<const float4*>(nbox + size_t(cl[j]) * 8u)[0];
        const float4 b1 = reinterpret_cast<const float4*>(nbox + size_t(cl[j]) * 8u)[1];
        const float dx = __builtin_fmaxf(a1.x, b1.x) - __builtin_fminf(a0.x, b0.x);
        const float dy = __builtin_fmaxf(a1.y, b1.y) - __builtin_fminf(a0.y, b0.y);
        const float dz = __builtin_fmaxf(a1.z, b1.z) - __builtin_fminf(a0.z, b0.z);
        const float area = dx * dy + dy * dz + dz * dx;
        if (area < best) best = area, bj = j;  // (ties: the lower position; the same value from both sides of a pair)
    }
    nn[i] = bj;
}

__global__ __launch_bounds__(kTpb) void ploc_merge(Work* w, const uint32_t* __restrict__ cl, const uint32_t* __restrict__ m_ptr,
                                                  uint32_t m_bound, const uint32_t* __restrict__ nn, Tree t, uint32_t n_leaves,
                                                  uint32_t* __restrict__ cl_out, uint32_t* __restrict__ valid) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    const uint32_t m = *m_ptr;
    if (i >= m) {
        if (i < m_bound) valid[i] = 0u;  // (the scan runs over the bound)
        return;
    }
    const uint32_t j = nn[i];
    uint32_t id = cl[i], keep = 1u;
    if (j != i && nn[j] == i) {  // a mutual pair: the lower position carries the merged cluster, the higher one disappears
        if (i < j) {
            const uint32_t a = cl[i], b = cl[j];
            id = n_leaves + atomicAdd(&w->n_internal, 1u);
            const float* pa = t.nbox + size_t(a) * 8u;
            const float* pb = t.nbox + size_t(b) * 8u;
            float* dst = t.nbox + size_t(id) * 8u;
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k) lo[k] = __builtin_fminf(pa[k], pb[k]), hi[k] = __builtin_fmaxf(pa[4 + k], pb[4 + k]);
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            dst[0] = lo[0], dst[1] = lo[1], dst[2] = lo[2], dst[3] = __builtin_fmaxf(pa[3], pb[3]);
            dst[4] = hi[0], dst[5] = hi[1], dst[6] = hi[2], dst[7] = dx * dy + dy * dz + dz * dx;
            t.left[id] = int32_t(a), t.right[id] = int32_t(b);
            t.size[id] = t.size[a] + t.size[b];
        } else {
            keep = 0u;
        }
    }
    cl_out[i] = id;
    valid[i] = keep;
}

__global__ __launch_bounds__(kTpb) void ploc_compact(const uint32_t* __restrict__ cl_in, const uint32_t* __restrict__ valid,
                                                    const uint32_t* __restrict__ pos, const uint32_t* __restrict__ m_ptr,
                                                    uint32_t* __restrict__ cl_next, uint32_t* m_next) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    const uint32_t m = *m_ptr;
    if (i >= m) return;
    if (valid[i]) cl_next[pos[i]] = cl_in[i];
    if (i == m - 1u) *m_next = pos[i] + valid[i];
}

__global__ void ploc_root(Work* w, const uint32_t* cl) { w->root = cl[0]; }
__global__ void ploc_seed(Work* w, uint32_t n) { w->ploc_m[0] = n; }
__global__ void radix_root(Work* w) { w->root = w->n_valid; }  // internal node 0

struct QItem {
    uint32_t bnode;   // binary node to turn into a 4-wide node
    uint32_t out;     // its index in the 4-wide node array
    uint32_t offset;  // position of its first triangle in the output order
};

// One level of the collapse. A candidate is a node of the binary tree; one with <= kLeafMax triangles below it becomes
// a leaf, a larger one may be opened into its two children (the left child's triangles come first in the output).
__global__ __launch_bounds__(kTpb) void collapse_level(Work* w, Tree t, uint32_t n_leaves, const uint32_t* __restrict__ sorted,
                                                       uint32_t* __restrict__ order, const QItem* __restrict__ q_in,
                                                       const uint32_t* __restrict__ q_in_count, QItem* __restrict__ q_out,
                                                       uint32_t* q_out_count, uint32_t* q_spare_count, BvhNode4* __restrict__ nodes,
                                                       uint32_t node_cap, uint32_t tri_base, uint32_t depth) {
    const uint32_t tid = blockIdx.x * kTpb + threadIdx.x;
    // (the levels of one batch are launched without the host in between: this level's count is read here, the launch is
    // sized by a bound; the third count word -- the level after next's output -- is zeroed here instead of by a
    // hipMemsetAsync per level, whose first use in a process loads the runtime's fill kernels)
    if (tid == 0) *q_spare_count = 0u;
    const uint32_t n_in = min(*q_in_count, node_cap);  // (overflow is flagged in w)
    if (tid >= n_in) return;
    const QItem it = q_in[tid];
    uint32_t c[4] = {uint32_t(t.left[it.bnode]), uint32_t(t.right[it.bnode]), 0u, 0u};
    uint32_t off[4] = {it.offset, it.offset + t.size[c[0]], 0u, 0u};
    int n = 2;
    auto expandable = [&](uint32_t id) { return t.size[id] > uint32_t(kLeafMax); };
    // open the expandable candidate with the largest surface (as bvh.cpp's collapse does); slots that are still free
    // when only leaves are left go to the halves of the largest leaf of two or more triangles (two tighter boxes
    // instead of one, for the same node)
    for (int pass = 0; pass < 2; ++pass)
        while (n < 4) {
            int pick = -1;
            float best = -1.0f;
            for (int k = 0; k < n; ++k)
                if (pass == 0 ? expandable(c[k]) : (t.size[c[k]] >= 2u && !expandable(c[k]))) {
                    const float a = t.nbox[size_t(c[k]) * 8u + 7u];
                    if (a > best) best = a, pick = k;
                }
            if (pick < 0) break;
            const uint32_t open = c[pick], o0 = off[pick];
            c[pick] = uint32_t(t.left[open]);
            c[n] = uint32_t(t.right[open]);
            off[n] = o0 + t.size[c[pick]];
            ++n;
        }
    BvhNode4 o;
    const float pinf = __uint_as_float(0x7f800000u), ninf = __uint_as_float(0xff800000u);
    for (int k = 0; k < 4; ++k) {
        if (k >= n) {  // (the empty box: bvh.cpp)
            o.lo_x[k] = o.lo_y[k] = o.lo_z[k] = pinf;
            o.hi_x[k] = o.hi_y[k] = o.hi_z[k] = ninf;
            o.child[k] = kNoChild;
            o.max_e12[k] = 0.0f;
            continue;
        }
        const float* v = t.nbox + size_t(c[k]) * 8u;
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
            q_out[atomicAdd(q_out_count, 1u)] = QItem{c[k], idx, off[k]};
            atomicMax(&w->max_depth, depth + 1u);
        } else {  // a leaf: its (<= kLeafMax) triangles go to positions off[k] .. of the output order
            uint32_t stack[2 * kLeafMax], sp = 0, cnt = 0;
            stack[sp++] = c[k];
            while (sp != 0) {
                const uint32_t id = stack[--sp];
                if (id < n_leaves) {
                    order[off[k] + cnt++] = sorted[id];
                } else {
                    stack[sp++] = uint32_t(t.right[id]);
                    stack[sp++] = uint32_t(t.left[id]);
                }
            }
            for (uint32_t a = 1; a < cnt; ++a) {  // ascending reference index inside a leaf (as bvh.cpp)
                const uint32_t x = order[off[k] + a];
                uint32_t b = a;
                while (b > 0 && order[off[k] + b - 1] > x) order[off[k] + b] = order[off[k] + b - 1], --b;
                order[off[k] + b] = x;
            }
            o.child[k] = ~int32_t(((tri_base + off[k]) << kLeafBits) | (cnt - 1u));
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

__global__ __launch_bounds__(kTpb) void iota_kernel(uint32_t* a, uint32_t n) {
    const uint32_t i = blockIdx.x * kTpb + threadIdx.x;
    if (i < n) a[i] = i;
}

__global__ void init_work(Work* w) {
    w->n_valid = 0;
    for (int k = 0; k < 3; ++k) w->cmin[k] = 0xFFFFFFFFu, w->cmax[k] = 0u;
    w->n_nodes = 1;  // the root
    w->overflow = 0;
    w->max_depth = 0;
    w->q_count[0] = w->q_count[1] = w->q_count[2] = 0;
    w->ploc_m[0] = w->ploc_m[1] = 0;
    w->max_e12 = 0.0f;
    w->n_internal = 0;
    w->root = 0;
}
// The tree's root: its pad for the mesh table, and the collapse's first candidate.
__global__ void root_info(Work* w, const float* nbox, QItem* q0) {
    w->max_e12 = nbox[size_t(w->root) * 8u + 3u];
    q0[0] = QItem{w->root, 0u, 0u};
    w->q_count[0] = 1u, w->q_count[1] = 0u, w->q_count[2] = 0u;
}

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
                            DeviceBvhResult* res, hipStream_t stream, int algo) {
    res->ok = false;
    res->d_nodes = nullptr;
    const uint32_t n_tested = (n_total / 8u) * 8u;  // triangle.rs:166-167
    if (n_tested < 8u) return hipSuccess;           // (callers route tiny meshes to the host builder)
    const Soa m = {soa.v0x, soa.v0y, soa.v0z, soa.e1x, soa.e1y, soa.e1z, soa.e2x, soa.e2y, soa.e2z, soa.is_padding};
    const uint32_t node_cap = n_tested / 2u + 8u;
    Work* w = nullptr;
    float* boxes = nullptr;
    uint64_t *keys = nullptr, *keys2 = nullptr;
    uint32_t *vals = nullptr, *sorted = nullptr, *order = nullptr, *arrived = nullptr;
    uint32_t *cl[2] = {nullptr, nullptr}, *cl_tmp = nullptr, *nn = nullptr, *valid = nullptr, *pos = nullptr;
    Tree t = {nullptr, nullptr, nullptr, nullptr, nullptr};
    QItem* queue[2] = {nullptr, nullptr};
    void *tmp = nullptr, *scan_tmp = nullptr;
    BvhNode4* nodes = nullptr;
    // ONE allocation for all the scratch (two dozen hipMalloc / hipFree pairs cost more than the kernels of a 70k-triangle
    // build): sized for n_valid = n_tested, carved up by a bump pointer; the node array, which outlives the build, apart.
    char* arena = nullptr;
    size_t arena_used = 0, arena_bytes = 0;
    auto cleanup = [&]() {
        (void)hipFree(arena);
        if (!res->ok) (void)hipFree(nodes);
    };
    size_t sort_bytes = 0, scan_bytes = 0;
    {
        hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys2, vals, sorted, n_total, 0, 64, stream);
        if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, scan_bytes, valid, pos, 0u, n_tested, rocprim::plus<uint32_t>(), stream);
        if (e != hipSuccess) return e;
    }
    {
        const size_t n_ids_max = 2u * size_t(n_tested);
        const auto r = [](size_t b) { return (b + 255u) & ~size_t(255); };
        arena_bytes = r(sizeof(Work)) + r(size_t(n_total) * 8u * sizeof(float)) + 2u * r(size_t(n_total) * sizeof(uint64_t)) +
                      2u * r(size_t(n_total) * sizeof(uint32_t)) + r(sort_bytes + 16u) + r(scan_bytes + 16u) +
                      r(n_ids_max * 8u * sizeof(float)) + 5u * r(n_ids_max * sizeof(uint32_t)) + 7u * r(size_t(n_tested) * sizeof(uint32_t)) +
                      2u * r(size_t(node_cap) * sizeof(QItem)) + r(sizeof(uint32_t)) + 4096u;
    }
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&arena), arena_bytes));
    auto alloc = [&](auto** p, size_t count) {
        const size_t bytes = (std::max<size_t>(count, 1) * sizeof(**p) + 255u) & ~size_t(255);
        if (arena_used + bytes > arena_bytes) return hipErrorOutOfMemory;  // (cannot happen: the sizes above bound every request)
        *p = reinterpret_cast<std::remove_reference_t<decltype(*p)>>(arena + arena_used);
        arena_used += bytes;
        return hipSuccess;
    };
    DEV_TRY(alloc(&w, 1));
    DEV_TRY(alloc(&boxes, size_t(n_total) * 8u));
    DEV_TRY(alloc(&keys, n_total));
    DEV_TRY(alloc(&keys2, n_total));
    DEV_TRY(alloc(&vals, n_total));
    DEV_TRY(alloc(&sorted, n_total));
    hipLaunchKernelGGL(init_work, dim3(1), dim3(1), 0, stream, w);
    hipLaunchKernelGGL(prim_setup, dim3(blocks(n_total)), dim3(kTpb), 0, stream, m, n_total, n_tested, boxes, w);
    hipLaunchKernelGGL(morton_keys, dim3(blocks(n_total)), dim3(kTpb), 0, stream, boxes, n_total, w, keys, vals);
    {
        char* tp = nullptr;
        DEV_TRY(alloc(&tp, sort_bytes + 16u));
        tmp = tp;
    }
    DEV_TRY(rocprim::radix_sort_pairs(tmp, sort_bytes, keys, keys2, vals, sorted, n_total, 0, 64, stream));
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
    const size_t n_ids = 2u * size_t(n) - 1u;
    DEV_TRY(alloc(&t.nbox, n_ids * 8u));
    DEV_TRY(alloc(&t.left, n_ids));
    DEV_TRY(alloc(&t.right, n_ids));
    DEV_TRY(alloc(&t.size, n_ids));
    DEV_TRY(alloc(&order, n));
    DEV_TRY(alloc(&queue[0], node_cap));
    DEV_TRY(alloc(&queue[1], node_cap));
    DEV_TRY(hipMalloc(reinterpret_cast<void**>(&nodes), size_t(node_cap) * sizeof(BvhNode4)));
    hipLaunchKernelGGL(init_leaves, dim3(blocks(n)), dim3(kTpb), 0, stream, w, sorted, boxes, t);
    bool built = false;
    if (algo == 0) {  // PLOC
        DEV_TRY(alloc(&cl[0], n));
        DEV_TRY(alloc(&cl[1], n));
        DEV_TRY(alloc(&cl_tmp, n));
        DEV_TRY(alloc(&nn, n));
        DEV_TRY(alloc(&valid, n));
        DEV_TRY(alloc(&pos, n));
        {
            char* tp = nullptr;
            DEV_TRY(alloc(&tp, scan_bytes + 16u));
            scan_tmp = tp;
        }
        hipLaunchKernelGGL(iota_kernel, dim3(blocks(n)), dim3(kTpb), 0, stream, cl[0], n);
        hipLaunchKernelGGL(ploc_seed, dim3(1), dim3(1), 0, stream, w, n);
        uint32_t* d_pm = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(w) + offsetof(Work, ploc_m));
        uint32_t mcur = n, radius = uint32_t(kPlocRadius);
        const char* lab = std::getenv("RBRT_HIP_LAB");  // (a lab knob: include/rbrt_hip_debug.h)
        if (const char* e = (lab && lab[0] == '1') ? std::getenv("RBRT_PLOC_RADIUS") : nullptr) radius = uint32_t(std::min(256, std::max(1, std::atoi(e))));
        int cur = 0, rounds = 0;
        bool stalled = false;
        // (a round typically merges ~45 % of the clusters; with many ties in merged surface -- coincident or duplicate
        // triangles, non-finite areas -- the lowest-position tie-break leaves one mutual pair per round: a round that
        // merges under 1 % of the clusters hands the build to the radix tree below instead of running hundreds more)
        // The host looks at the count once per BATCH of rounds (a look is a 40-us round trip, a round's four launches take
        // less): half the rounds the count is expected to need at 45 % per round, three at least; a batch's launches are
        // sized by the count at its start.
        while (mcur > 1u && rounds < 400 && !stalled) {
            const int batch = std::min(400 - rounds, std::max(3, int(std::ceil(std::log(double(mcur)) / 0.6 * 0.5))));
            const uint32_t before = mcur;
            for (int r = 0; r < batch; ++r) {
                const uint32_t* m_in = d_pm + (rounds & 1);
                uint32_t* m_out = d_pm + ((rounds & 1) ^ 1);
                hipLaunchKernelGGL(ploc_nearest, dim3(blocks(before)), dim3(kTpb), 0, stream, cl[cur], m_in, t.nbox, nn, radius);
                hipLaunchKernelGGL(ploc_merge, dim3(blocks(before)), dim3(kTpb), 0, stream, w, cl[cur], m_in, before, nn, t, n, cl_tmp, valid);
                DEV_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, valid, pos, 0u, before, rocprim::plus<uint32_t>(), stream));
                hipLaunchKernelGGL(ploc_compact, dim3(blocks(before)), dim3(kTpb), 0, stream, cl_tmp, valid, pos, m_in, cl[cur ^ 1], m_out);
                cur ^= 1;
                ++rounds;
            }
            DEV_TRY(hipMemcpyAsync(&mcur, d_pm + (rounds & 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            DEV_TRY(hipStreamSynchronize(stream));
            // (under 1 % per round, compounded over the batch)
            stalled = before > 256u && double(mcur) > double(before) * std::pow(0.99, double(batch));
        }
        if (mcur == 1u) {
            hipLaunchKernelGGL(ploc_root, dim3(1), dim3(1), 0, stream, w, cl[cur]);
            built = true;
        }
    }
    if (!built) {  // binary radix tree (also the fallback if the clustering did not converge)
        DEV_TRY(alloc(&t.parent, n_ids));
        DEV_TRY(alloc(&arrived, n_ids));
        DEV_TRY(hipMemsetAsync(arrived, 0, n_ids * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(radix_tree, dim3(blocks(n - 1)), dim3(kTpb), 0, stream, keys2, w, t);
        hipLaunchKernelGGL(fit, dim3(blocks(n)), dim3(kTpb), 0, stream, w, t, arrived);
        hipLaunchKernelGGL(radix_root, dim3(1), dim3(1), 0, stream, w);
    }
    hipLaunchKernelGGL(root_info, dim3(1), dim3(1), 0, stream, w, t.nbox, queue[0]);
    // collapse, level by level: the queue size comes back to the host once per four levels (a level has at most four times
    // the candidates of the one before: that sizes the launches in between)
    uint32_t n_in = 1, depth = 0;
    uint32_t* d_q_count = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(w) + offsetof(Work, q_count));
    while (n_in != 0 && depth <= uint32_t(kMaxBvhDepth)) {
        uint64_t bound = n_in;
        for (int k = 0; k < 4 && depth <= uint32_t(kMaxBvhDepth); ++k, ++depth, bound *= 4u) {
            const uint32_t qi = depth % 3u, qo = (depth + 1u) % 3u, qs = (depth + 2u) % 3u;
            hipLaunchKernelGGL(collapse_level, dim3(blocks(uint32_t(std::min<uint64_t>(bound, node_cap)))), dim3(kTpb), 0, stream, w, t, n, sorted,
                               order, queue[depth & 1u], d_q_count + qi, queue[(depth & 1u) ^ 1u], d_q_count + qo, d_q_count + qs, nodes,
                               node_cap, tri_base, depth);
        }
        DEV_TRY(hipMemcpyAsync(&n_in, d_q_count + depth % 3u, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        DEV_TRY(hipStreamSynchronize(stream));
        if (n_in > node_cap) n_in = node_cap;  // (overflow is flagged in w)
    }
    hipLaunchKernelGGL(emit_tris, dim3(blocks(n)), dim3(kTpb), 0, stream, m, w, order, d_tris_out);
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
