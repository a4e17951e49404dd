// bvh.cpp — binned-SAH builder (host): a binary tree first, then collapsed into 4-wide nodes.
// See bvh.h for what the structure must guarantee.
#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <array>
#include <atomic>
#include <functional>
#include <limits>
#include <thread>

namespace rbrt {
namespace {

constexpr int kBins = 16;
// SAH constants. On the GPU a node visit costs a dependent ~1 us fetch, a triangle test ~60 VALU
// instructions, so leaves are allowed to fill up (<= kLeafMax) before another level is added.
float cost_traverse() {
    static const float v = [] {
        const char* lab = std::getenv("RBRT_HIP_LAB");  // (a lab knob: include/rbrt_hip_debug.h)
        const char* e = (lab && lab[0] == '1') ? std::getenv("RBRT_BVH_CT") : nullptr;
        float x = e ? float(std::atof(e)) : 4.0f;
        return x > 0.0f ? x : 4.0f;
    }();
    return v;
}
constexpr float kCostTri = 1.0f;
// Binary inner nodes live at depths 0..kMaxInnerDepth; collapsing never deepens a path, so the
// 4-wide tree is at most that deep too.
constexpr int kMaxInnerDepth = kMaxBvhDepth;

// Binary node of the intermediate tree (both child boxes in the parent).
struct Node2 {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int32_t child0, child1;
    float max_e12_0, max_e12_1;
};

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<float>::max();
            hi[k] = -std::numeric_limits<float>::max();
        }
    }
    void grow(const float* p) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void grow(const Box& b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box box;
    float c[3];
    float e12;
    uint32_t idx;  // reference index (what the leaves are ordered by, what the scan's tie rule compares)
    uint32_t src;  // where the triangle's nine values are: position in the mesh's SoA arrays, or in the record array
};

struct ChildInfo {
    int32_t ref;
    Box box;
    float max_e12;
};

// What a (sub)tree build appends to: binary nodes in DFS pre-order, leaf records in DFS leaf order.
struct Sink {
    std::vector<Node2> nodes2;
    std::vector<BvhTri> tris;
    uint32_t n_leaves = 0;
};

// Subtrees below the top of the tree are built by worker threads into sinks of their own and spliced back
// in DFS order, which reproduces the single-threaded numbering exactly (same arrays for any thread count).
constexpr int32_t kTaskRefBase = INT32_MIN + 1;  // child refs kTaskRefBase + k stand for task k until the splice
inline bool is_task_ref(int32_t r) { return r != kNoChild && r < -(int32_t(1) << 30) - 1; }

// Spatial splits (Stich, Friedrich, Dietrich 2009, "Spatial Splits in Bounding Volume Hierarchies"). A long, thin or
// diagonal triangle -- a fin, a spike, the seam of a fold -- makes every box that holds it large, and an object split can
// only choose which side gets the whole of it. A spatial split cuts the node's box with a plane and gives each side a
// REFERENCE to the triangle with the box of the part on that side: the same record is then in two leaves. That is legal
// here because the leaf rounds merge with a lexicographic (t, index) minimum -- the same triangle found twice is the same
// key -- and because the parts' boxes cover the triangle: the point where the fp32 test accepts a hit lies (to within the
// traversal's pad) in at least one part, and the walk reaches that part's leaf (bvh.h). Duplicated references are capped
// at kSpatialBudget of the mesh's triangles, dealt down the tree in proportion to the subtrees' sizes, so that the tree is
// a function of the SET of triangles (any thread count, any order of the input).
constexpr size_t kThreadedFrom = 4096;  // entries from which the subtrees are built by worker threads
constexpr size_t kSlicedFrom = 32768;   // references from which a node of the TOP of the tree is binned in slices by helper threads
constexpr int kSpatialBins = 32;
constexpr int kSpatialLostRun = 5;
constexpr float kSpatialAlpha = 1e-4f;  // try a spatial split when the object split's children overlap by more than this share of the root's surface
float spatial_budget_frac() {
    static const float v = [] {
        const char* lab = std::getenv("RBRT_HIP_LAB");  // (a lab knob: include/rbrt_hip_debug.h)
        const char* e = (lab && lab[0] == '1') ? std::getenv("RBRT_BVH_SPATIAL") : nullptr;
        const float x = e ? float(std::atof(e)) : kSpatialBudget;
        return x >= 0.0f && x <= kSpatialBudget ? x : kSpatialBudget;  // (the record array is sized for kSpatialBudget)
    }();
    return v;
}

struct Builder {
    const rbrt_mesh_t* m = nullptr;  // the source: a mesh's SoA arrays ...
    const BvhTri* recs = nullptr;    // ... or n_recs triangle records in any order
    size_t n_recs = 0;
    BvhBuildOptions opt;
    std::vector<Node2> nodes2;
    BvhBuildResult out;
    float root_area = 0.0f;
    struct Task {
        std::vector<Prim> refs;
        int depth;
        int64_t budget;
        int lost;
        Sink sink;
        ChildInfo root;
    };
    std::vector<Task> tasks;
    size_t task_grain = 0;  // 0: single-threaded build, no tasks

    bool cancelled() const { return opt.cancel && opt.cancel->load(std::memory_order_relaxed); }

    void corners(uint32_t src, float v[3][3]) const {  // the triangle as the boxes see it: v0, v0 + e1, v0 + e2 in float
        float e1[3], e2[3];
        if (recs) {
            const BvhTri& r = recs[src];
            v[0][0] = r.v0[0], v[0][1] = r.v0[1], v[0][2] = r.v0[2];
            e1[0] = r.e1x, e1[1] = r.e1yz[0], e1[2] = r.e1yz[1];
            e2[0] = r.e2xy[0], e2[1] = r.e2xy[1], e2[2] = r.e2z;
        } else {
            v[0][0] = m->v0x[src], v[0][1] = m->v0y[src], v[0][2] = m->v0z[src];
            e1[0] = m->e1x[src], e1[1] = m->e1y[src], e1[2] = m->e1z[src];
            e2[0] = m->e2x[src], e2[1] = m->e2y[src], e2[2] = m->e2z[src];
        }
        const float fmax = std::numeric_limits<float>::max();
        for (int k = 0; k < 3; ++k) {  // clamp in case v0 + e overflows
            v[1][k] = std::min(std::max(v[0][k] + e1[k], -fmax), fmax);
            v[2][k] = std::min(std::max(v[0][k] + e2[k], -fmax), fmax);
        }
    }

    BvhTri make_tri(const Prim& p) const {
        BvhTri t;
        if (recs) {
            t = recs[p.src];
        } else {
            const uint32_t i = p.src;
            t.v0[0] = m->v0x[i], t.v0[1] = m->v0y[i], t.v0[2] = m->v0z[i];
            t.e1x = m->e1x[i], t.e1yz[0] = m->e1y[i], t.e1yz[1] = m->e1z[i];
            t.e2xy[0] = m->e2x[i], t.e2xy[1] = m->e2y[i], t.e2z = m->e2z[i];
        }
        t.index = p.idx;
        t.pad[0] = t.pad[1] = 0;
        return t;
    }

    // How many triangles fit under a subtree whose root inner node would sit at `depth`.
    static uint64_t capacity(int depth) {
        if (depth > kMaxInnerDepth) return kLeafMax;
        return uint64_t(kLeafMax) << (kMaxInnerDepth - depth + 1);
    }

    ChildInfo make_leaf(Sink& sk, std::vector<Prim>& refs, size_t count, const Box& box, float max_e12) {
        uint32_t first = uint32_t(sk.tris.size());
        // deterministic order inside a leaf: ascending reference index
        std::sort(refs.begin(), refs.begin() + count, [](const Prim& x, const Prim& y) { return x.idx < y.idx; });
        for (size_t i = 0; i < count; ++i) sk.tris.push_back(make_tri(refs[i]));
        ++sk.n_leaves;
        return ChildInfo{~int32_t((first << kLeafBits) | (uint32_t(count) - 1u)), box, max_e12};
    }

    // The boxes of the two parts of reference p's triangle on either side of the plane x[axis] = plane, each inside p's
    // own box (which earlier cuts may have made smaller than the triangle's). The split coordinate of a crossing point is
    // the plane itself, so the two parts meet exactly; the other two coordinates of a crossing point are widened by a few
    // units in the last place: a part's box may be a little too large, never too small.
    void split_ref(const Prim& p, int axis, float plane, Box& lb, Box& rb) const {
        float v[3][3];
        corners(p.src, v);
        split_tri(v, p.box, axis, plane, lb, rb);
    }
    // (the same with the triangle's corners in hand: a reference chopped into many bins fetches them once)
    static void split_tri(const float v[3][3], const Box& pbox, int axis, float plane, Box& lb, Box& rb) {
        lb.reset(), rb.reset();
        for (int i = 0; i < 3; ++i) {
            const float* a = v[i];
            const float* b = v[(i + 1) % 3];
            if (a[axis] <= plane) lb.grow(a);
            if (a[axis] >= plane) rb.grow(a);
            if ((a[axis] < plane && b[axis] > plane) || (a[axis] > plane && b[axis] < plane)) {
                const float t = (plane - a[axis]) / (b[axis] - a[axis]);
                float lo[3], hi[3];
                for (int k = 0; k < 3; ++k) {
                    const float x = a[k] + (b[k] - a[k]) * t;
                    const float slack = 4.0f * 1.1920929e-7f * (std::fabs(a[k]) + std::fabs(b[k])) + 1e-30f;
                    lo[k] = std::min(std::max(x - slack, std::min(a[k], b[k])), std::max(a[k], b[k]));
                    hi[k] = std::max(std::min(x + slack, std::max(a[k], b[k])), std::min(a[k], b[k]));
                }
                lo[axis] = hi[axis] = plane;
                lb.grow(lo), lb.grow(hi), rb.grow(lo), rb.grow(hi);
            }
        }
        lb.hi[axis] = std::min(lb.hi[axis], plane), rb.lo[axis] = std::max(rb.lo[axis], plane);
        for (int k = 0; k < 3; ++k) {  // inside the reference's own box
            lb.lo[k] = std::max(lb.lo[k], pbox.lo[k]), lb.hi[k] = std::min(lb.hi[k], pbox.hi[k]);
            rb.lo[k] = std::max(rb.lo[k], pbox.lo[k]), rb.hi[k] = std::min(rb.hi[k], pbox.hi[k]);
        }
    }
    static bool box_ok(const Box& b) { return b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2]; }
    // A reference that spans bins b0 < b1 of `axis` (planes[j] = lower end of bin j): the box of the triangle's part in
    // each of them -- bin b0 stands for everything below planes[b0 + 1], bin b1 for everything above planes[b1] --, inside
    // the reference's own box, grown into bin_box. One walk over the corners and over the planes every edge crosses
    // (a crossing point as in split_tri: on the plane exactly, widened by a few units in the last place elsewhere), where
    // cutting the triangle plane by plane touched all three edges per plane.
    static void chop_tri(const float v[3][3], const Box& pbox, int axis, const float* planes, int b0, int b1, Box* bin_box) {
        Box part[kSpatialBins];
        for (int b = b0; b <= b1; ++b) part[b].reset();
        for (int i = 0; i < 3; ++i) {
            const float x = v[i][axis];
            int b = b0;
            while (b < b1 && x > planes[b + 1]) ++b;  // planes[b] <= x <= planes[b + 1] (open-ended at b0 and b1)
            part[b].grow(v[i]);
            if (b < b1 && x == planes[b + 1]) part[b + 1].grow(v[i]);  // (a corner ON a plane belongs to both sides)
        }
        for (int i = 0; i < 3; ++i) {
            const float* a = v[i];
            const float* c = v[(i + 1) % 3];
            const float xa = a[axis], xc = c[axis];
            if (xa == xc) continue;
            const float xlo = std::min(xa, xc), xhi = std::max(xa, xc);
            const float inv = 1.0f / (xc - xa);
            float slack[3], mn[3], mx[3];
            for (int k = 0; k < 3; ++k) {
                slack[k] = 4.0f * 1.1920929e-7f * (std::fabs(a[k]) + std::fabs(c[k])) + 1e-30f;
                mn[k] = std::min(a[k], c[k]), mx[k] = std::max(a[k], c[k]);
            }
            for (int j = b0 + 1; j <= b1; ++j) {
                const float plane = planes[j];
                if (!(xlo < plane && plane < xhi)) continue;
                const float t = (plane - xa) * inv;
                float lo[3], hi[3];
                for (int k = 0; k < 3; ++k) {
                    const float x = a[k] + (c[k] - a[k]) * t;
                    lo[k] = std::min(std::max(x - slack[k], mn[k]), mx[k]);
                    hi[k] = std::max(std::min(x + slack[k], mx[k]), mn[k]);
                }
                lo[axis] = hi[axis] = plane;
                part[j - 1].grow(lo), part[j - 1].grow(hi), part[j].grow(lo), part[j].grow(hi);
            }
        }
        for (int b = b0; b <= b1; ++b) {
            Box& q = part[b];
            if (b > b0) q.lo[axis] = std::max(q.lo[axis], planes[b]);
            if (b < b1) q.hi[axis] = std::min(q.hi[axis], planes[b + 1]);
            for (int k = 0; k < 3; ++k) q.lo[k] = std::max(q.lo[k], pbox.lo[k]), q.hi[k] = std::min(q.hi[k], pbox.hi[k]);
            if (box_ok(q)) bin_box[b].grow(q);
        }
    }
    static void set_centroid(Prim& p) {
        for (int k = 0; k < 3; ++k) p.c[k] = 0.5f * p.box.lo[k] + 0.5f * p.box.hi[k];
    }

    // fn(slice, begin, end) over [0, n) in `slices` contiguous slices, slice 0 on this thread, the others on threads of their own.
    unsigned slice_threads = 1;
    template <class F>
    static void in_slices(size_t n, unsigned slices, F fn) {
        if (slices <= 1u) {
            fn(0u, size_t(0), n);
            return;
        }
        const size_t per = (n + slices - 1) / slices;
        std::vector<std::thread> th;
        for (unsigned sl = 1; sl < slices; ++sl) th.emplace_back([&fn, sl, per, n] { fn(sl, std::min(n, sl * per), std::min(n, (sl + 1) * per)); });
        fn(0u, size_t(0), std::min(n, per));
        for (auto& t : th) t.join();
    }

    // Every reference of `refs` placed by `place(p, left, right)` (which appends to one list or to both), the lists in the
    // order of the references: in slices by helper threads, the slices' lists joined in their order.
    template <class F>
    static void two_lists(const std::vector<Prim>& refs, unsigned slices, size_t n_left, size_t n_right, F place, std::vector<Prim>& left,
                          std::vector<Prim>& right) {
        if (slices <= 1u) {
            left.reserve(n_left), right.reserve(n_right);
            for (const Prim& p : refs) place(p, left, right);
            return;
        }
        std::vector<std::vector<Prim>> ls(slices), rs(slices);
        in_slices(refs.size(), slices, [&](unsigned sl, size_t i0, size_t i1) {
            ls[sl].reserve((i1 - i0) / 2 + 16), rs[sl].reserve((i1 - i0) / 2 + 16);
            for (size_t i = i0; i < i1; ++i) place(refs[i], ls[sl], rs[sl]);
        });
        size_t nl = 0, nr = 0;
        for (unsigned sl = 0; sl < slices; ++sl) nl += ls[sl].size(), nr += rs[sl].size();
        left.reserve(nl), right.reserve(nr);
        for (unsigned sl = 0; sl < slices; ++sl) {
            left.insert(left.end(), ls[sl].begin(), ls[sl].end()), right.insert(right.end(), rs[sl].begin(), rs[sl].end());
            std::vector<Prim>().swap(ls[sl]), std::vector<Prim>().swap(rs[sl]);
        }
    }

    // `top`: this call belongs to the single-threaded top of the tree; lists of at most task_grain references are not
    // built here but queued as tasks (a placeholder ref is returned). `budget`: references this subtree may still add.
    ChildInfo build_node(Sink& sk, std::vector<Prim> refs, int depth, bool top, int64_t budget, int lost = 0) {
        const size_t count = refs.size();
        if (top && count <= task_grain) {
            tasks.push_back(Task{std::move(refs), depth, budget, lost, Sink(), ChildInfo()});
            ChildInfo c;
            c.ref = kTaskRefBase + int32_t(tasks.size() - 1);
            c.box.reset();
            c.max_e12 = 0.0f;
            return c;  // box and max_e12 are filled in by the splice
        }
        Box box, cbox;
        box.reset();
        cbox.reset();
        float max_e12 = 0.0f;
        for (const Prim& p : refs) {
            box.grow(p.box);
            cbox.grow(p.c);
            max_e12 = std::max(max_e12, p.e12);
        }
        if (depth > kMaxInnerDepth || count <= 1) return make_leaf(sk, refs, count, box, max_e12);
        if (cancelled()) return make_leaf(sk, refs, 1, box, max_e12);  // (memory-safe nonsense: the result is thrown away)

        // ---- object split: binned SAH over the three axes (centroids of the references' boxes) ----
        // (both binnings are sums over the references of boxes grown and counts: at the top of a large tree, where one
        // thread works alone, they are done in slices by helper threads and added up -- the same bins whatever the slicing)
        const unsigned slices = top && count >= kSlicedFrom ? std::min<unsigned>(slice_threads, unsigned(count / 4096)) : 1u;
        const float parent_area = box.half_area();
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_split = -1;
        Box best_lbox, best_rbox;
        best_lbox.reset(), best_rbox.reset();
        {
            struct ObjBins {
                Box box[3][kBins];
                uint32_t cnt[3][kBins];
            };
            float cmin[3], scale[3];
            bool axis_ok[3];
            for (int axis = 0; axis < 3; ++axis) {
                const float cext = cbox.hi[axis] - cbox.lo[axis];
                axis_ok[axis] = cext > 0.0f;
                cmin[axis] = cbox.lo[axis], scale[axis] = axis_ok[axis] ? float(kBins) / cext : 0.0f;
            }
            std::vector<ObjBins> part(slices);
            in_slices(count, slices, [&](unsigned sl, size_t i0, size_t i1) {
                ObjBins& B = part[sl];
                for (int axis = 0; axis < 3; ++axis)
                    for (int i = 0; i < kBins; ++i) B.box[axis][i].reset(), B.cnt[axis][i] = 0;
                for (size_t i = i0; i < i1; ++i) {
                    const Prim& p = refs[i];
                    for (int axis = 0; axis < 3; ++axis) {
                        if (!axis_ok[axis]) continue;
                        const int bi = std::min(kBins - 1, std::max(0, int((p.c[axis] - cmin[axis]) * scale[axis])));
                        B.box[axis][bi].grow(p.box);
                        ++B.cnt[axis][bi];
                    }
                }
            });
            ObjBins& B = part[0];
            for (unsigned sl = 1; sl < slices; ++sl)
                for (int axis = 0; axis < 3; ++axis)
                    for (int i = 0; i < kBins; ++i) B.box[axis][i].grow(part[sl].box[axis][i]), B.cnt[axis][i] += part[sl].cnt[axis][i];
            for (int axis = 0; axis < 3; ++axis) {
                if (!axis_ok[axis]) continue;
                const Box* bin_box = B.box[axis];
                const uint32_t* bin_cnt = B.cnt[axis];
                Box right_box[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                acc.reset();
                uint32_t n = 0;
                for (int i = kBins - 1; i > 0; --i) {
                    acc.grow(bin_box[i]);
                    n += bin_cnt[i];
                    right_box[i] = acc;
                    right_cnt[i] = n;
                }
                acc.reset();
                n = 0;
                for (int i = 1; i < kBins; ++i) {  // split = first bin of the right side
                    acc.grow(bin_box[i - 1]);
                    n += bin_cnt[i - 1];
                    if (n == 0 || right_cnt[i] == 0) continue;
                    float cost = acc.half_area() * float(n) + right_box[i].half_area() * float(right_cnt[i]);
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = axis;
                        best_split = i;
                        best_lbox = acc, best_rbox = right_box[i];
                    }
                }
            }
        }
        // ---- spatial split: only where the object split leaves its children overlapping, and while references may be added ----
        float sp_cost = std::numeric_limits<float>::infinity(), sp_plane = 0.0f;
        int sp_axis = -1;
        Box sp_lbox, sp_rbox;
        uint32_t sp_nl = 0, sp_nr = 0;
        sp_lbox.reset(), sp_rbox.reset();
        // (`lost`: how many nodes in a row, up the path to this one, tried a spatial split and kept the object split: after
        // kSpatialLostRun of them the subtree is taken to be one where cutting triangles does not pay -- a property of the path,
        // so of the set of triangles. A smooth mesh is then binned spatially on its top levels only: build 90 -> 58 ms for the
        // bunny stand-in with the SAH cost within 0.05 %, the rough stand-in's tree unchanged)
        bool sp_tried = false;
        if (lost < kSpatialLostRun && budget > 0 && count > size_t(kLeafMax) && best_axis >= 0 && parent_area > 0.0f && root_area > 0.0f) {
            Box ov;
            for (int k = 0; k < 3; ++k) ov.lo[k] = std::max(best_lbox.lo[k], best_rbox.lo[k]), ov.hi[k] = std::min(best_lbox.hi[k], best_rbox.hi[k]);
            if (box_ok(ov) && ov.half_area() > kSpatialAlpha * root_area) {
                sp_tried = true;
                struct SpBins {
                    Box box[3][kSpatialBins];
                    uint32_t enter[3][kSpatialBins], leave[3][kSpatialBins];
                };
                float lo3[3], scale[3], planes[3][kSpatialBins + 1];
                bool axis_ok[3];
                for (int axis = 0; axis < 3; ++axis) {
                    const float lo = box.lo[axis], ext = box.hi[axis] - box.lo[axis];
                    axis_ok[axis] = ext > 0.0f && std::isfinite(ext);
                    lo3[axis] = lo, scale[axis] = axis_ok[axis] ? float(kSpatialBins) / ext : 0.0f;
                    for (int j = 0; j <= kSpatialBins; ++j) planes[axis][j] = j == kSpatialBins ? box.hi[axis] : lo + ext * (float(j) / float(kSpatialBins));
                }
                std::vector<SpBins> part(slices);
                in_slices(count, slices, [&](unsigned sl, size_t i0, size_t i1) {
                    SpBins& S = part[sl];
                    for (int axis = 0; axis < 3; ++axis)
                        for (int i = 0; i < kSpatialBins; ++i) S.box[axis][i].reset(), S.enter[axis][i] = S.leave[axis][i] = 0;
                    for (size_t i = i0; i < i1; ++i) {
                        const Prim& p = refs[i];
                        float v[3][3];
                        bool have_v = false;
                        for (int axis = 0; axis < 3; ++axis) {
                            if (!axis_ok[axis]) continue;
                            const float* pl = planes[axis];
                            int b0 = std::min(kSpatialBins - 1, std::max(0, int((p.box.lo[axis] - lo3[axis]) * scale[axis])));
                            int b1 = std::min(kSpatialBins - 1, std::max(b0, int((p.box.hi[axis] - lo3[axis]) * scale[axis])));
                            while (b1 > b0 && p.box.hi[axis] <= pl[b1]) --b1;      // (a box that ends ON a plane is not beyond it)
                            while (b0 < b1 && p.box.lo[axis] >= pl[b0 + 1]) ++b0;
                            ++S.enter[axis][b0], ++S.leave[axis][b1];
                            if (b0 == b1) {
                                S.box[axis][b0].grow(p.box);
                            } else {  // chopped into the bins it crosses
                                if (!have_v) corners(p.src, v), have_v = true;
                                chop_tri(v, p.box, axis, pl, b0, b1, S.box[axis]);
                            }
                        }
                    }
                });
                SpBins& S = part[0];
                for (unsigned sl = 1; sl < slices; ++sl)
                    for (int axis = 0; axis < 3; ++axis)
                        for (int i = 0; i < kSpatialBins; ++i) {
                            S.box[axis][i].grow(part[sl].box[axis][i]);
                            S.enter[axis][i] += part[sl].enter[axis][i], S.leave[axis][i] += part[sl].leave[axis][i];
                        }
                for (int axis = 0; axis < 3; ++axis) {
                    if (!axis_ok[axis]) continue;
                    const Box* bin_box = S.box[axis];
                    const uint32_t *enter = S.enter[axis], *leave = S.leave[axis];
                    const float* pl = planes[axis];
                    Box right_box[kSpatialBins];
                    uint32_t right_cnt[kSpatialBins];
                    Box acc;
                    acc.reset();
                    uint32_t n = 0;
                    for (int i = kSpatialBins - 1; i > 0; --i) {
                        acc.grow(bin_box[i]);
                        n += leave[i];
                        right_box[i] = acc, right_cnt[i] = n;
                    }
                    acc.reset();
                    n = 0;
                    for (int i = 1; i < kSpatialBins; ++i) {  // the plane between bins i - 1 and i
                        acc.grow(bin_box[i - 1]);
                        n += enter[i - 1];
                        if (n == 0 || right_cnt[i] == 0 || n == count || right_cnt[i] == count) continue;
                        const float cost = acc.half_area() * float(n) + right_box[i].half_area() * float(right_cnt[i]);
                        if (cost < sp_cost) sp_cost = cost, sp_axis = axis, sp_plane = pl[i], sp_lbox = acc, sp_rbox = right_box[i], sp_nl = n, sp_nr = right_cnt[i];
                    }
                }
            }
        }
        const float leaf_cost = kCostTri * float(count);
        float split_cost = std::numeric_limits<float>::infinity();
        if (best_axis >= 0 && parent_area > 0.0f) split_cost = cost_traverse() + kCostTri * std::min(best_cost, sp_cost) / parent_area;
        if (count <= size_t(kLeafMax) && leaf_cost <= split_cost) return make_leaf(sk, refs, count, box, max_e12);

        std::vector<Prim> left, right;
        const uint64_t cap = capacity(depth + 1);
        int64_t used = 0;
        if (sp_axis >= 0 && sp_cost < best_cost) {
            // ---- partition by the plane; a reference that crosses it goes to both sides with the boxes of its parts, unless
            // keeping it whole on one side is cheaper (judged against the sides the sweep found: the same for every order) ----
            const float a_l = sp_lbox.half_area(), a_r = sp_rbox.half_area();
            const auto place = [&](const Prim& p, std::vector<Prim>& left, std::vector<Prim>& right) {
                if (p.box.hi[sp_axis] <= sp_plane) {
                    left.push_back(p);
                } else if (p.box.lo[sp_axis] >= sp_plane) {
                    right.push_back(p);
                } else {
                    Box lb, rb;
                    split_ref(p, sp_axis, sp_plane, lb, rb);
                    const bool l_ok = box_ok(lb), r_ok = box_ok(rb);
                    Box wl = sp_lbox, wr = sp_rbox;
                    wl.grow(p.box), wr.grow(p.box);
                    const float c_split = a_l * float(sp_nl) + a_r * float(sp_nr);
                    const float c_left = wl.half_area() * float(sp_nl) + a_r * float(sp_nr - 1u);
                    const float c_right = a_l * float(sp_nl - 1u) + wr.half_area() * float(sp_nr);
                    if (!r_ok || (l_ok && c_left <= c_split && c_left <= c_right)) {
                        left.push_back(p);
                    } else if (!l_ok || (c_right <= c_split)) {
                        right.push_back(p);
                    } else {
                        Prim pl = p, pr = p;
                        pl.box = lb, pr.box = rb;
                        set_centroid(pl), set_centroid(pr);
                        left.push_back(pl), right.push_back(pr);
                    }
                }
            };
            two_lists(refs, slices, sp_nl, sp_nr, place, left, right);
            used = int64_t(left.size() + right.size()) - int64_t(count);
            if (left.empty() || right.empty() || left.size() == count || right.size() == count || used > budget || left.size() > cap ||
                right.size() > cap)
                left.clear(), right.clear(), used = 0;  // (no gain, or beyond the budget: the object split after all)
        }
        if (left.empty()) {
            size_t mid = 0;
            if (best_axis >= 0) {
                const float cmin = cbox.lo[best_axis];
                const float scale = float(kBins) / (cbox.hi[best_axis] - cbox.lo[best_axis]);
                const auto place = [&](const Prim& p, std::vector<Prim>& l, std::vector<Prim>& r) {
                    const int bi = std::min(kBins - 1, std::max(0, int((p.c[best_axis] - cmin) * scale)));
                    (bi < best_split ? l : r).push_back(p);
                };
                two_lists(refs, slices, 0, 0, place, left, right);
                mid = left.size();
                if (mid == 0 || mid == count || mid > cap || (count - mid) > cap) left.clear(), right.clear();
            }
            if (left.empty()) {
                // degenerate or too unbalanced for the depth budget: median split along the widest axis
                int axis = 0;
                for (int k = 1; k < 3; ++k)
                    if (cbox.hi[k] - cbox.lo[k] > cbox.hi[axis] - cbox.lo[axis]) axis = k;
                mid = count / 2;
                std::nth_element(refs.begin(), refs.begin() + mid, refs.end(), [axis](const Prim& x, const Prim& y) {
                    return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.idx < y.idx);
                });
                left.assign(refs.begin(), refs.begin() + mid);
                right.assign(refs.begin() + mid, refs.end());
            }
        }
        std::vector<Prim>().swap(refs);  // (the list is in its two halves now)
        const int64_t rest = std::max<int64_t>(0, budget - used);
        // (dealt by the surface of the references' boxes, not by their number: the triangles that gain from being cut are the
        // large ones, and a fin of a few hundred of them would get next to nothing of a budget dealt by count)
        double w_l = 0.0, w_r = 0.0;
        for (const Prim& p : left) w_l += double(p.box.half_area());
        for (const Prim& p : right) w_r += double(p.box.half_area());
        const int64_t b_left = w_l + w_r > 0.0 ? int64_t(double(rest) * (w_l / (w_l + w_r))) : rest / 2;
        const uint32_t node = uint32_t(sk.nodes2.size());
        sk.nodes2.emplace_back();
        const int lost_next = used > 0 ? 0 : sp_tried ? lost + 1 : lost;
        ChildInfo l = build_node(sk, std::move(left), depth + 1, top, b_left, lost_next);
        ChildInfo r = build_node(sk, std::move(right), depth + 1, top, rest - b_left, lost_next);
        set_node(sk.nodes2, node, l, r);
        return ChildInfo{int32_t(node), box, max_e12};
    }

    static void set_node(std::vector<Node2>& nodes, uint32_t node, const ChildInfo& l, const ChildInfo& r) {
        Node2& n = nodes[node];
        for (int k = 0; k < 3; ++k) {
            n.lo0[k] = l.box.lo[k], n.hi0[k] = l.box.hi[k];
            n.lo1[k] = r.box.lo[k], n.hi1[k] = r.box.hi[k];
        }
        n.child0 = l.ref, n.child1 = r.ref;
        n.max_e12_0 = l.max_e12, n.max_e12_1 = r.max_e12;
    }

    static int32_t relocate(int32_t ref, uint32_t node_base, uint32_t tri_base) {
        if (ref >= 0) return int32_t(uint32_t(ref) + node_base);
        const uint32_t leaf = uint32_t(~ref);
        return ~int32_t((((leaf >> kLeafBits) + tri_base) << kLeafBits) | (leaf & uint32_t(kLeafMax - 1)));
    }

    // Appends a finished task's nodes and leaf records to the final arrays; returns its root as the parent sees it.
    ChildInfo splice_task(Task& t) {
        const uint32_t node_base = uint32_t(nodes2.size()), tri_base = uint32_t(out.tris.size());
        for (Node2 n : t.sink.nodes2) {
            n.child0 = relocate(n.child0, node_base, tri_base);
            n.child1 = relocate(n.child1, node_base, tri_base);
            nodes2.push_back(n);
        }
        out.tris.insert(out.tris.end(), t.sink.tris.begin(), t.sink.tris.end());
        out.n_leaves += t.sink.n_leaves;
        ChildInfo c = t.root;
        c.ref = relocate(c.ref, node_base, tri_base);
        t.sink = Sink();  // free early
        return c;
    }

    // DFS over the single-threaded top of the tree, in the order the single-threaded build numbers nodes.
    ChildInfo splice(const Sink& top, int32_t ref, const Box& box, float max_e12) {
        if (is_task_ref(ref)) return splice_task(tasks[size_t(ref - kTaskRefBase)]);
        const Node2 src = top.nodes2[size_t(ref)];
        const uint32_t me = uint32_t(nodes2.size());
        nodes2.emplace_back();
        Box b0, b1;
        for (int k = 0; k < 3; ++k) {
            b0.lo[k] = src.lo0[k], b0.hi[k] = src.hi0[k];
            b1.lo[k] = src.lo1[k], b1.hi[k] = src.hi1[k];
        }
        const ChildInfo l = splice(top, src.child0, b0, src.max_e12_0);
        const ChildInfo r = splice(top, src.child1, b1, src.max_e12_1);
        set_node(nodes2, me, l, r);
        return ChildInfo{int32_t(me), box, max_e12};
    }

    ChildInfo build_tree(std::vector<Prim> prims) {
        unsigned n_threads = std::thread::hardware_concurrency();
        if (opt.max_threads != 0) n_threads = std::min(n_threads, opt.max_threads);
        if (const char* e = std::getenv("RBRT_BVH_THREADS")) n_threads = unsigned(std::max(1, std::atoi(e)));
        n_threads = std::min(n_threads, 16u);
        const size_t n_prims = prims.size();
        const int64_t budget = int64_t(double(spatial_budget_frac()) * double(n_prims));
        {
            Box all;
            all.reset();
            for (const Prim& p : prims) all.grow(p.box);
            root_area = all.half_area();
        }
        if (n_threads <= 1 || n_prims < kThreadedFrom) {
            Sink sk;
            sk.nodes2.reserve(n_prims / 2 + 2);
            sk.tris.reserve(n_prims + size_t(budget) + 2);
            const ChildInfo root = build_node(sk, std::move(prims), 0, false, budget);
            nodes2 = std::move(sk.nodes2);
            out.tris = std::move(sk.tris);
            out.n_leaves = sk.n_leaves;
            return root;
        }
        // (the top of the tree, down to lists of task_grain references, is built by this thread alone: about sixteen tasks for a
        // small mesh, lists of 4096 or an eighth of a thread's share for a large one)
        task_grain = std::max<size_t>(std::min<size_t>(std::max<size_t>(n_prims / 16u, 512), 4096), n_prims / (8u * n_threads));
        slice_threads = n_threads;
        Sink top;
        const ChildInfo top_root = build_node(top, std::move(prims), 0, true, budget);
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t k = next.fetch_add(1); k < tasks.size(); k = next.fetch_add(1)) {
                Task& t = tasks[k];
                t.root = build_node(t.sink, std::move(t.refs), t.depth, false, t.budget, t.lost);  // disjoint lists: no sharing
            }
        };
        std::vector<std::thread> pool;
        for (unsigned i = 1; i < n_threads; ++i) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
        task_grain = 0;
        if (cancelled()) return top_root;  // (what a cancelled build leaves is not a tree: nothing of it is walked, run() drops it)
        nodes2.reserve(n_prims / 2 + 2);
        out.tris.reserve(n_prims + size_t(budget) + 2);
        return splice(top, top_root.ref, top_root.box, top_root.max_e12);
    }

    // A leaf holding one zero-area triangle: |a| = 0 < eps rejects it for every ray
    // (triangle.rs:198-200), so it stands in for "no child".
    ChildInfo make_dummy() {
        uint32_t first = uint32_t(out.tris.size());  // (after the splice: out.tris is final)
        BvhTri t;
        std::memset(&t, 0, sizeof(t));
        t.index = 0xFFFFFFFFu;
        out.tris.push_back(t);
        Box b;
        for (int k = 0; k < 3; ++k) b.lo[k] = b.hi[k] = 0.0f;
        return ChildInfo{~int32_t((first << kLeafBits) | 0), b, 0.0f};
    }

    void run() {
        const uint32_t n_tested = recs ? uint32_t(n_recs) : (m->n_total / 8u) * 8u;  // triangle.rs:166-167
        std::vector<Prim> prims;
        prims.reserve(n_tested);
        for (uint32_t i = 0; i < n_tested; ++i) {
            if (recs ? recs[i].index == 0xFFFFFFFFu : (m->is_padding && m->is_padding[i])) continue;  // triangle.rs:400
            Prim p;
            float v0[3], e1[3], e2[3];
            if (recs) {
                const BvhTri& r = recs[i];
                v0[0] = r.v0[0], v0[1] = r.v0[1], v0[2] = r.v0[2];
                e1[0] = r.e1x, e1[1] = r.e1yz[0], e1[2] = r.e1yz[1];
                e2[0] = r.e2xy[0], e2[1] = r.e2xy[1], e2[2] = r.e2z;
            } else {
                v0[0] = m->v0x[i], v0[1] = m->v0y[i], v0[2] = m->v0z[i];
                e1[0] = m->e1x[i], e1[1] = m->e1y[i], e1[2] = m->e1z[i];
                e2[0] = m->e2x[i], e2[1] = m->e2y[i], e2[2] = m->e2z[i];
            }
            // Non-finite v0/e1/e2 can never pass the ordered compares of triangle.rs:198-241
            // (a, u, v or t comes out inf/NaN), so such a triangle is not indexed.
            bool finite = true;
            for (int k = 0; k < 3; ++k)
                finite = finite && std::isfinite(v0[k]) && std::isfinite(e1[k]) && std::isfinite(e2[k]);
            if (!finite) continue;
            const float fmax = std::numeric_limits<float>::max();
            float v1[3], v2[3];
            for (int k = 0; k < 3; ++k) {  // clamp in case v0 + e overflows
                v1[k] = std::min(std::max(v0[k] + e1[k], -fmax), fmax);
                v2[k] = std::min(std::max(v0[k] + e2[k], -fmax), fmax);
            }
            p.box.reset();
            p.box.grow(v0), p.box.grow(v1), p.box.grow(v2);
            for (int k = 0; k < 3; ++k) p.c[k] = 0.5f * p.box.lo[k] + 0.5f * p.box.hi[k];
            float l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
            float l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
            p.e12 = l1 * l2;
            p.idx = recs ? recs[i].index : i;
            p.src = i;
            prims.push_back(p);
        }
        out.n_indexed = uint32_t(prims.size());
        if (prims.empty()) {
            nodes2.emplace_back();
            ChildInfo d0 = make_dummy();
            set_node(nodes2, 0, d0, d0);
        } else {
            ChildInfo root = build_tree(std::move(prims));
            if (cancelled()) {
                out.cancelled = true;
                return;
            }
            if (root.ref < 0) {  // whole mesh fits one leaf
                nodes2.emplace_back();
                ChildInfo d = make_dummy();
                set_node(nodes2, 0, root, d);
            }
            out.max_e12 = root.max_e12;
        }
        if (cancelled()) {
            out.cancelled = true;
            return;
        }
        out.nodes.reserve(nodes2.size() / 2 + 2);
        plan_openings();
        collapse(0, 0);
        out.stack_need = 3u * (out.max_depth + 1u) + 1u;
        nodes2.clear();
        nodes2.shrink_to_fit();
    }

    // The openings that minimise the 4-wide tree's SAH cost for THIS binary tree (dynamic programme over slots, after
    // Ylitie et al. 2017 section 3.1): E[n][k] = cheapest way to stand for binary subtree n with at most k slots of one
    // wide node; C[n] = n as a wide node of its own (one visit of its surface + its four slots' best use).
    // The triangles of a leaf with their boxes (each cut to the leaf's box: a reference cut by a spatial split lies in it).
    struct LeafTris {
        uint32_t first, count;
        BvhTri t[kLeafMax];
        Box tb[kLeafMax];
        float te12[kLeafMax];
    };
    void leaf_tris(int32_t ref, const Box& leaf_box, LeafTris& L) const {
        const uint32_t code = uint32_t(~ref);
        L.first = code >> kLeafBits, L.count = (code & uint32_t(kLeafMax - 1)) + 1u;
        const float fmax = std::numeric_limits<float>::max();
        for (uint32_t i = 0; i < L.count; ++i) {
            const BvhTri& t = L.t[i] = out.tris[L.first + i];
            const float v0[3] = {t.v0[0], t.v0[1], t.v0[2]};
            const float e1[3] = {t.e1x, t.e1yz[0], t.e1yz[1]}, e2[3] = {t.e2xy[0], t.e2xy[1], t.e2z};
            float v1[3], v2[3];
            for (int k = 0; k < 3; ++k) {
                v1[k] = std::min(std::max(v0[k] + e1[k], -fmax), fmax);
                v2[k] = std::min(std::max(v0[k] + e2[k], -fmax), fmax);
            }
            Box& b = L.tb[i];
            b.reset();
            b.grow(v0), b.grow(v1), b.grow(v2);
            for (int k = 0; k < 3; ++k) b.lo[k] = std::max(b.lo[k], leaf_box.lo[k]), b.hi[k] = std::min(b.hi[k], leaf_box.hi[k]);
            L.te12[i] = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) *
                        std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        }
    }
    // The cheapest way to stand for a leaf with at most k leaves (sum of surface x count over the groups of a partition of
    // its triangles): label[i] = group of triangle i, groups numbered in order of their first triangle. Returns the cost
    // in units of surface x triangles and the number of groups.
    // (every partition of `count` triangles into at most four groups, in the order of their label strings; a group is a
    // bit mask of triangles)
    struct Partition {
        uint8_t label[kLeafMax], groups, mask[4];
    };
    static const std::vector<Partition>& partitions_of(uint32_t count) {
        static const std::array<std::vector<Partition>, kLeafMax + 1> all = [] {
            std::array<std::vector<Partition>, kLeafMax + 1> t;
            for (uint32_t c = 1; c <= uint32_t(kLeafMax); ++c) {
                Partition cur{};
                std::function<void(uint32_t, int)> rec = [&](uint32_t i, int used) {
                    if (i == c) {
                        cur.groups = uint8_t(used);
                        for (int g = 0; g < 4; ++g) cur.mask[g] = 0;
                        for (uint32_t j = 0; j < c; ++j) cur.mask[cur.label[j]] |= uint8_t(1u << j);
                        t[c].push_back(cur);
                        return;
                    }
                    for (int g = 0; g <= used && g < 4; ++g) cur.label[i] = uint8_t(g), rec(i + 1, std::max(used, g + 1));
                };
                rec(0, 0);
            }
            return t;
        }();
        return all[count];
    }
    static float best_partition(const LeafTris& L, int k, uint8_t label[kLeafMax], int& groups) {
        float sub[1u << kLeafMax];  // surface x count of every subset of the leaf's triangles
        for (uint32_t m = 1; m < (1u << L.count); ++m) {
            Box b;
            b.reset();
            uint32_t n = 0;
            for (uint32_t j = 0; j < L.count; ++j)
                if (m >> j & 1u) b.grow(L.tb[j]), ++n;
            sub[m] = b.half_area() * float(n);
        }
        float best = std::numeric_limits<float>::infinity();
        const Partition* pick = nullptr;
        for (const Partition& q : partitions_of(L.count)) {
            if (int(q.groups) > k) continue;
            float cost = 0.0f;
            for (int g = 0; g < int(q.groups); ++g) cost += sub[q.mask[g]];
            if (cost < best) best = cost, pick = &q;
        }
        groups = pick ? int(pick->groups) : 1;  // (no finite cost: one group)
        for (uint32_t j = 0; j < L.count; ++j) label[j] = pick ? pick->label[j] : uint8_t(0);
        return best;
    }

    std::vector<std::array<uint8_t, 5>> dp_open;  // [binary node][k] = slots given to child 0 (0: the node stays closed)
    std::vector<uint8_t> dp_open_root;            // [binary node] = slots given to child 0 when the node IS a wide node
    void plan_openings() {
        const size_t nn = nodes2.size();
        // (one wide visit = 1; a triangle of a leaf 0.2: measured on the GPU, 0.03-0.2 are equal, 0.35 costs the rough
        // stand-in 1 %, 1.0 costs it 6 %: profiles/r05_collapse_ab.txt)
        const float c_node = 1.0f, c_tri = 0.2f;
        std::vector<std::array<float, 5>> E(nn);
        dp_open.assign(nn, std::array<uint8_t, 5>{});
        dp_open_root.assign(nn, 1);
        std::vector<float> area(nn, 0.0f);
        {
            Box rb;
            rb.reset();
            const Node2& r = nodes2[0];
            for (int k = 0; k < 3; ++k)
                rb.lo[k] = std::min(r.lo0[k], r.lo1[k]), rb.hi[k] = std::max(r.hi0[k], r.hi1[k]);
            area[0] = rb.half_area();
        }
        for (size_t i = 0; i < nn; ++i) {  // (children come after their parents)
            const Node2& n = nodes2[i];
            Box a, b;
            for (int k = 0; k < 3; ++k) a.lo[k] = n.lo0[k], a.hi[k] = n.hi0[k], b.lo[k] = n.lo1[k], b.hi[k] = n.hi1[k];
            if (n.child0 >= 0) area[size_t(n.child0)] = a.half_area();
            if (n.child1 >= 0) area[size_t(n.child1)] = b.half_area();
        }
        auto leaf_table = [&](int32_t ref, const Box& box, float e[5]) {
            LeafTris L;
            leaf_tris(ref, box, L);
            e[1] = box.half_area() * c_tri * float(L.count);
            uint8_t label[kLeafMax];
            int groups;
            for (int k = 2; k <= 4; ++k) e[k] = L.count > 1 ? std::min(e[k - 1], c_tri * best_partition(L, k, label, groups)) : e[1];
        };
        for (size_t ii = nn; ii-- > 0;) {
            const Node2& n = nodes2[ii];
            float el[5], er[5];
            {
                Box a, b;
                for (int k = 0; k < 3; ++k) a.lo[k] = n.lo0[k], a.hi[k] = n.hi0[k], b.lo[k] = n.lo1[k], b.hi[k] = n.hi1[k];
                if (n.child0 < 0) leaf_table(n.child0, a, el);
                if (n.child1 < 0) leaf_table(n.child1, b, er);
                for (int k = 1; k <= 4; ++k) {
                    if (n.child0 >= 0) el[k] = E[size_t(n.child0)][k];
                    if (n.child1 >= 0) er[k] = E[size_t(n.child1)][k];
                }
            }
            float D[5];
            uint8_t Dl[5];
            for (int k = 2; k <= 4; ++k) {
                D[k] = std::numeric_limits<float>::infinity(), Dl[k] = 1;
                for (int l = 1; l < k; ++l)
                    if (el[l] + er[k - l] < D[k]) D[k] = el[l] + er[k - l], Dl[k] = uint8_t(l);
            }
            const float C = area[ii] * c_node + D[4];
            dp_open_root[ii] = Dl[4];
            E[ii][1] = C;
            for (int k = 2; k <= 4; ++k) {
                if (D[k] < C)
                    E[ii][k] = D[k], dp_open[ii][k] = Dl[k];
                else
                    E[ii][k] = C, dp_open[ii][k] = 0;
            }
        }
    }

    // Binary -> 4-wide by the table of plan_openings().
    uint32_t collapse(int32_t n2, uint32_t depth) {
        struct Cand {
            int32_t ref;
            Box box;
            float e12;
        };
        auto children_of = [&](int32_t idx, Cand& a, Cand& b) {
            const Node2& n = nodes2[size_t(idx)];
            a.ref = n.child0, b.ref = n.child1;
            a.e12 = n.max_e12_0, b.e12 = n.max_e12_1;
            for (int k = 0; k < 3; ++k) {
                a.box.lo[k] = n.lo0[k], a.box.hi[k] = n.hi0[k];
                b.box.lo[k] = n.lo1[k], b.box.hi[k] = n.hi1[k];
            }
        };
        Cand c[4];
        int n = 0;
        // The openings the table chose for this node's four slots: a closed binary node becomes a wide node of its own, a
        // leaf given several slots is regrouped in place into that many leaves (each keeps the ascending order of
        // reference indices).
        std::function<void(const Cand&, int)> gather = [&](const Cand& x, int k) {
            if (x.ref < 0 && k >= 2 && (uint32_t(~x.ref) & uint32_t(kLeafMax - 1)) != 0u) {
                LeafTris L;
                leaf_tris(x.ref, x.box, L);
                uint8_t label[kLeafMax];
                int groups = 1;
                const float cost = best_partition(L, k, label, groups);
                if (groups > 1 && cost < x.box.half_area() * float(L.count)) {
                    uint32_t pos = L.first;
                    for (int g = 0; g < groups; ++g) {
                        Cand& d = c[n++];
                        d.box.reset(), d.e12 = 0.0f;
                        const uint32_t start = pos;
                        for (uint32_t j = 0; j < L.count; ++j)
                            if (label[j] == g) out.tris[pos++] = L.t[j], d.box.grow(L.tb[j]), d.e12 = std::max(d.e12, L.te12[j]);
                        d.ref = ~int32_t((start << kLeafBits) | (pos - start - 1u));
                    }
                    out.n_leaves += uint32_t(groups - 1);
                    return;
                }
            }
            const int left = x.ref >= 0 && k >= 2 ? dp_open[size_t(x.ref)][size_t(k)] : 0;
            if (left == 0) {
                c[n++] = x;
                return;
            }
            Cand a, b;
            children_of(x.ref, a, b);
            gather(a, left), gather(b, k - left);
        };
        {
            Cand a, b;
            children_of(n2, a, b);
            const int left = dp_open_root[size_t(n2)];
            gather(a, left), gather(b, 4 - left);
        }
        const uint32_t me = uint32_t(out.nodes.size());
        out.nodes.emplace_back();
        out.max_depth = std::max(out.max_depth, depth);
        int32_t refs[4];
        for (int i = 0; i < n; ++i) refs[i] = c[i].ref >= 0 ? int32_t(collapse(c[i].ref, depth + 1)) : c[i].ref;
        BvhNode4& o = out.nodes[me];
        // (an unused slot holds the EMPTY box, lo = +inf / hi = -inf: every ray's entry parameter to it is +inf and its
        // exit parameter -inf, so the traversal's single compare max(tn, eps) <= min(tf, best) fails: kernels.hip child_key)
        const float pinf = std::numeric_limits<float>::infinity(), ninf = -pinf;
        for (int i = 0; i < 4; ++i) {
            const bool used = i < n;
            o.lo_x[i] = used ? c[i].box.lo[0] : pinf, o.lo_y[i] = used ? c[i].box.lo[1] : pinf;
            o.lo_z[i] = used ? c[i].box.lo[2] : pinf, o.hi_x[i] = used ? c[i].box.hi[0] : ninf;
            o.hi_y[i] = used ? c[i].box.hi[1] : ninf, o.hi_z[i] = used ? c[i].box.hi[2] : ninf;
            o.child[i] = used ? refs[i] : kNoChild;
            o.max_e12[i] = used ? c[i].e12 : 0.0f;
        }
        return me;
    }
};

}  // namespace

BvhBuildResult build_bvh(const rbrt_mesh_t& mesh, const BvhBuildOptions& opt) {
    Builder b;
    b.m = &mesh, b.opt = opt;
    b.run();
    return std::move(b.out);
}

BvhBuildResult build_bvh_from_records(const BvhTri* recs, size_t n, const BvhBuildOptions& opt) {
    Builder b;
    static const BvhTri none{};
    b.recs = recs ? recs : &none, b.n_recs = recs ? n : 0, b.opt = opt;
    b.run();
    return std::move(b.out);
}

}  // namespace rbrt
