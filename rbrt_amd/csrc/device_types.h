// device_types.h — HBM-resident data layout shared by the host-side builder and the gfx950 kernels.
#pragma once
#include <stdint.h>

#include "../../include/rbrt_hip.h"

namespace rbrt {

// Compile-time limits.
// Tile number -> tile coordinates and back (rbrt_hip.h "How tiles are dealt to ranks": row ty is rotated by RBRT_TILE_SKEW * ty).
#if defined(__HIPCC__)
#define RBRT_HOST_DEVICE __host__ __device__
#else
#define RBRT_HOST_DEVICE  // (bvh.cpp alone under a host compiler: tests/cpp)
#endif
RBRT_HOST_DEVICE inline void tile_xy(uint32_t tile, uint32_t tiles_x, uint32_t& ty, uint32_t& tx) {
    ty = tile / tiles_x;
    tx = uint32_t((uint64_t(tile - ty * tiles_x) + uint64_t(RBRT_TILE_SKEW) * ty) % tiles_x);
}
RBRT_HOST_DEVICE inline uint32_t tile_number(uint32_t ty, uint32_t tx, uint32_t tiles_x) {
    const uint32_t rot = uint32_t((uint64_t(RBRT_TILE_SKEW) * ty) % tiles_x);
    return ty * tiles_x + (tx + tiles_x - rot) % tiles_x;
}
constexpr uint32_t kTileListHeader = 4;  // words ahead of the lists in TraceParams::tile_lists
constexpr int kBlock = 256;          // threads per workgroup of the test hooks' kernels
// Threads per workgroup of the short kernels that run BESIDE resident trace launches (resolve, sky_resolve, unpack, the tile
// pass): ONE wave. The persistent trace waves hold every wave slot and all of every CU's LDS until they exit, one by one; a
// 256-thread workgroup needs FOUR free slots on one CU at once, and a kernel on a higher-priority stream whose workgroups do
// not fit makes the dispatcher keep what frees up for it instead of handing it to the next trace launch's waves. Round 4's
// kernel trace (profiles/r04_trace_frames.txt): the resolve -- 50 us of work -- took 3.0-3.4 ms in EVERY frame of a stream,
// its lane waiting for it all the while. A one-wave workgroup runs in any slot a trace wave leaves.
#ifndef RBRT_SMALL_BLOCK
#define RBRT_SMALL_BLOCK 64
#endif
constexpr int kSmallBlock = RBRT_SMALL_BLOCK;
constexpr int kMaxBvhDepth = 20;     // deepest 4-wide node (root = 0) the builder may create
constexpr int kStackMax = 3 * (kMaxBvhDepth + 1) + 1;  // a visit defers at most 3 children per level
constexpr int kMaxPathDepth = 64;    // opts.max_depth limit (reference: 50, lib.rs:99)
constexpr int kMaxObjects = 255;     // spheres + meshes (object id is stored in one byte per bounce)
#ifndef RBRT_LEAF_BITS
#define RBRT_LEAF_BITS 2
#endif
constexpr int kLeafBits = RBRT_LEAF_BITS;  // width of the count field of a leaf link
constexpr int kLeafMax = 1 << kLeafBits;   // triangles per leaf
constexpr int kPoolMax = 256;        // largest path pool per wave the persistent megakernel is built for
constexpr uint32_t kWorkShards = 8;         // work-item counters (one per XCD)
constexpr uint32_t kWorkCounterStride = 16;  // in u64: each counter on its own 128-B line
constexpr uint32_t kMaxShadeRounds = 64;  // hard bound of register-resident shading rounds per pass
constexpr int kLdsStack = 8;         // per-lane traversal stack entries kept in LDS by the megakernel; deeper
                                     // entries spill (exactly) to a per-wave global scratch

// One 4-wide BVH node = 128 B = one cache line, fetched as 8 x dwordx4 by ONE lane: the four child boxes
// as SoA (so the four slab tests are the same code on four registers), four child links and, per
// child, max |e1|*|e2| of its subtree (the error-bound term of the culling pad).
// child >= 0 : index of an inner node; child < 0 : leaf, ~child = (first_tri << kLeafBits) | (count - 1);
// kNoChild marks an unused slot, whose box is EMPTY (lo = +inf, hi = -inf) so that the slab test fails for every ray.
constexpr int32_t kNoChild = INT32_MIN;
struct alignas(128) BvhNode4 {
    float lo_x[4], lo_y[4], lo_z[4];
    float hi_x[4], hi_y[4], hi_z[4];
    int32_t child[4];
    float max_e12[4];
};
static_assert(sizeof(BvhNode4) == 128, "node must be 128 B");
// the traversal addresses planes by byte offset: lo_x 0, lo_y 16, lo_z 32, hi_x 48, hi_y 64, hi_z 80, child 96, max_e12 112

// One triangle record = 48 B (3 x dwordx4), stored in leaf order. The 9 fp32 values are exactly
// the 9 SoA streams the reference kernel reads (triangle.rs:177-187); `index` is the triangle's
// position in the reference arrays (ties in t resolve to the lowest index: triangle.rs:400).
struct alignas(16) BvhTri {
    float v0[3];
    float e1x;
    float e1yz[2];
    float e2xy[2];
    float e2z;
    uint32_t index;
    uint32_t pad[2];
};
static_assert(sizeof(BvhTri) == 48, "tri must be 48 B");

struct alignas(16) Normal4 {
    float x, y, z, w;
};

struct DevSphere {
    float center[3];
    float radius;
};

// A BasicTriangle element (triangle.rs:9-34): first corner, the two edges from it, the stored normal.
struct DevTriangle {
    float v0[3], e0[3], e1[3], normal[3];
};

struct DevMaterial {
    float albedo[3];
    float param;
    int32_t kind;
};

struct DevMesh {
    const BvhNode4* nodes;
    const BvhTri* tris;      // the SCENE's triangle array (all meshes; this mesh's leaves point into its own part)
    const Normal4* normals;  // [reference index] -> (nx, ny, nz, 0)
    float bbox_lo[3];
    float bbox_hi[3];
    float center[3];  // of the bbox
    float radius;     // half diagonal of the bbox
    float max_e12;    // max |e1|*|e2| over the indexed triangles
    uint32_t n_nodes;
    uint32_t n_tris;
};

struct DevCounters {  // mirrors rbrt_hip_stats_t's counters
    unsigned long long rays, mesh_gate_pass, nodes_visited, tris_tested, mesh_hits, samples,
        nan_discriminants;
    // megakernel diagnostics (counting variant only; rbrt_hip_scene_debug_counters):
    // [0..5] passes per status kind, [6..11] lanes used per kind, [12] traversal wave-steps,
    // [13] active lane-steps, [14] refill rounds, [15] census rounds
    // [16] cycles in traversal steps, [17] cycles in shading passes, [18] total cycles (s_memtime, summed over waves)
    // [19] leaf rounds, [20] lanes with a leaf in those rounds, [21] node-walk rounds, [22] lanes walking in them
    // [23] waves that gave up on a bounded wait (must stay 0)
    // [24] min wave start, [25] max time work ran out, [26] max wave end, [27] sum of (end - start) (s_memrealtime, 100 MHz)
    unsigned long long diag[64];
};

// Kernel arguments of one trace launch (passed by value).
struct TraceParams {
    rbrt_camera_t cam;
    float min_dist, max_dist, eps_frac;  // eps_frac = 1/min_dist (triangle.rs:146)
    float bg[3];
    uint32_t max_depth;
    uint64_t seed_key;  // splitmix64(seed)
    uint32_t n_spheres, n_meshes;
    uint32_t n_elem_tris;          // BasicTriangle elements (rbrt_scene_t::triangles)
    const DevSphere* spheres;
    const DevTriangle* elem_tris;
    const uint32_t* elems;         // [n_spheres + n_elem_tris] Scene::elements order: bit 31 set = elem_tris[low bits], else spheres[..];
                                   // null when there are no triangles (element e is sphere e)
    const DevMaterial* materials;  // [object id]: the elements in their order, then the meshes
    const DevMesh* meshes;
    const BvhTri* tris;  // the triangle records of ALL meshes, one array; leaf links hold absolute positions in it
    // work decomposition
    uint32_t tiles_x, tiles_y, n_tiles;
    uint32_t tile_rank, tile_world, n_local_tiles;
    uint32_t sample_base;   // first sample index of this batch
    uint32_t batch;         // samples in this batch
    uint32_t tiles_reversed;  // 1: tiles are handed out last-to-first (scheduling only, see api.cpp empty_end_is_first)
    uint32_t batch_magic, tiles_x_magic;  // floor(2^32 / d) (d = 1: 2^32 - 1) for div_magic() in the kernels
    uint64_t n_items;       // n_local_tiles * 64 * batch
    float* sample_buf;      // [batch][n_local_tiles*64][3]
    DevCounters* counters;
    // persistent megakernel only
    unsigned long long* work_counter;  // [kWorkShards * kWorkCounterStride] next unclaimed item per shard (zeroed before every launch)
    uint32_t* gseq;                    // [n_waves][kPoolMax][kMaxPathDepth/4] scatter records beyond the 4 kept in LDS
    uint32_t stack_entries;            // per-lane traversal stack entries kept in LDS (<= stack_need)
    uint32_t* gstack;                  // [n_waves][kStackMax][64] overflow of the LDS stacks
    uint32_t y_low_water;              // refill a traversal pass when fewer lanes than this are busy
    uint32_t y_high_water, y_high_min_parked;  // ... or fewer than y_high_water while at least that many rays are parked
    uint32_t leaf_round;               // test deferred leaves once this many lanes are stalled on one ...
    uint32_t share_idle;               // shared traversals: idle lanes needed for a round of giving (0: never)
    uint32_t leaf_leaves;              // ... or once this many leaves are pending (a leaf round deals their triangles out to all lanes)
    uint32_t work_stripes;             // chunks per stripe when the work shards interleave over the item range (0: contiguous eighths)
    uint32_t drain_mode;               // scheduling once the work items have run out (bits: megakernel.inl "drain")
    uint32_t shade_rounds;             // shading pass: rounds a sphere-only bounce chain may stay in registers
    uint32_t shade_cont_min;           // ... as long as at least this many lanes continue (ignored once the work has run out)
    // The tile pass (kernels.hip primary_cull_kernel + tile_lists_kernel, run before the trace launch; both null: off).
    // tile_cull: one word per tile of the image: bit e < 24 = no camera ray of the tile can reach element e (a sphere),
    // bit 24 + m (m < 7) = none passes mesh m's box, bit 31 = all of them and the scene has nothing else: the tile sees
    // the background only. tile_lists: [0] = n_work, [1] = n_sky, then from [kTileListHeader] the rank's n_work local
    // tile numbers the trace kernel renders, ascending, and from [kTileListHeader + n_local_tiles] the n_sky
    // background-only ones, which sky_resolve_kernel finishes without the trace kernel ever seeing them.
    uint32_t* tile_cull;
    uint32_t* tile_lists;
    uint32_t tile_list_mode;  // order of the work list (tile_lists_kernel)
    uint32_t tile_tail_div;   // mode 4: the last n_work / this light tiles of the row-major order are handed out at the very end
    uint32_t tile_lists_wide;  // 1: tile_lists_kernel as four waves (a launch that has the GPU to itself); 0: as one wave
    // Helper launches (api.cpp "Elastic launches"): a second launch of the kernel that joins THIS launch's work -- same work
    // counters, same sample buffer -- when the GPU has room for more waves than the launch was issued with.
    // helper_words[0] = helper waves that may be holding work of the lane's launch, [1] = sequence number of the lane's
    // last launch that has been resolved (its counters are the next launch's by now); helper_seq = this launch's number.
    uint32_t* helper_words;
    uint32_t helper_seq;
    uint32_t wave_base;  // helper launches: first per-wave scratch slot (gseq / gstack) of this launch's waves; 0 for the launch itself
    uint32_t helper_min_items;  // helper launches: a helper wave joins only while at least this many work items per wave (the launch's and all its helpers') are left
};

struct ResolveParams {
    uint32_t width, height, tiles_x, n_tiles;
    uint32_t tile_rank, tile_world, n_local_tiles;
    uint32_t batch;
    uint32_t first_batch, last_batch;  // flags
    float inv_spp;                     // 1.0f / spp (lib.rs:101)
    const float* sample_buf;
    float* acc;           // [n_local_tiles*64][3] running sum
    float* out_radiance;  // row-major image if tile_world <= 1, else packed tiles; may be null
    uint8_t* out_rgb8;    // same indexing; may be null
    unsigned long long* work_counter;  // the finished launch's work counters, zeroed here for the lane's next launch
    const uint32_t* tile_lists;        // TraceParams::tile_lists (null: every local tile has samples in sample_buf)
    DevCounters* counters;             // sky_resolve_kernel in a counting launch: its samples and rays; else null
    uint32_t* helper_words;            // TraceParams::helper_words of the launch this resolves (null: none)
    uint32_t helper_seq;               // ... and its sequence number, published in helper_words[1]
    unsigned long long* error_flag;    // DevCounters::diag[57]: raised if helper waves of the launch never finish (bounded wait)
};

}  // namespace rbrt
