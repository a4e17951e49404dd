// bvh.h — host-side BVH builder (binned-SAH binary tree, collapsed to 4-wide nodes) for one triangle mesh.
//
// The reference has NO acceleration structure beyond one box per mesh (mesh.rs:232-243 tests every
// triangle for every ray that passes aabbox.rs:28-58). This BVH is purely a culling structure: the
// device traversal must return exactly what the brute-force scan of triangle.rs:134-262 +
// triangle.rs:392-410 returns (smallest accepted t, lowest reference index on ties). To that end
//   - only triangles the scan can ever return are indexed: i < 8*floor(n_total/8)
//     (chunks_exact(8), triangle.rs:166-167) and !is_padding[i] (triangle.rs:400);
//   - every node carries max |e1|*|e2| of its subtree, from which the traversal derives a per-ray
//     inflation of the node boxes that covers the fp32 error of the Moller-Trumbore test
//     (DESIGN.md "Why the BVH cannot change the answer").
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "device_types.h"

namespace rbrt {

// The host builder may reference a triangle from more than one leaf (spatial splits, bvh.cpp): at most kSpatialBudget
// duplicated references per indexed triangle. A mesh of n_total entries never needs more than bvh_record_capacity records
// (the scan-visible entries, their duplicates, one dummy record of a degenerate tree).
constexpr float kSpatialBudget = 0.60f;
inline uint64_t bvh_record_capacity(uint32_t n_total) {
    const uint64_t n_tested = uint64_t(n_total / 8u) * 8u;
    return n_tested + uint64_t(double(kSpatialBudget) * double(n_tested)) + 2u;
}

struct BvhBuildResult {
    std::vector<BvhNode4> nodes;  // nodes[0] = root
    std::vector<BvhTri> tris;     // leaf order (a triangle cut by spatial splits appears once per leaf that references it)
    float max_e12 = 0.0f;
    uint32_t max_depth = 0;   // depth of the deepest 4-wide node (root = 0)
    uint32_t stack_need = 1;  // traversal stack entries that can ever be live: 3 per level + 1
    uint32_t n_indexed = 0;   // triangles in the BVH (tris.size() - n_indexed - [dummy] = duplicated references)
    uint32_t n_leaves = 0;
    bool cancelled = false;   // BvhBuildOptions::cancel was raised: the arrays are not a tree of the mesh
};

struct BvhBuildOptions {
    unsigned max_threads = 0;                    // 0: every hardware thread (at most 16; $RBRT_BVH_THREADS overrides)
    const std::atomic<bool>* cancel = nullptr;  // polled while building: a build nobody waits for any more ends early
};

// Builds over the mesh's SoA arrays (host pointers).
BvhBuildResult build_bvh(const rbrt_mesh_t& mesh, const BvhBuildOptions& opt = BvhBuildOptions());

// Builds over n triangle records in ANY order (`index` = the triangle's position in the reference arrays; records whose
// index is 0xFFFFFFFF are skipped): for the indexed records of a mesh -- e.g. read back from the device builder's output --
// this is, array for array, the tree build_bvh gives for that mesh (every decision of the build is a function of SETS of
// triangles; leaves are ordered by reference index).
BvhBuildResult build_bvh_from_records(const BvhTri* recs, size_t n, const BvhBuildOptions& opt = BvhBuildOptions());

}  // namespace rbrt
