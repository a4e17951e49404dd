// megakernel.inl — the persistent path-tracing megakernel (included by kernels.hip).
//
// One wave64 per workgroup, a pool of POOLN path slots in LDS, and wavefront ballots to sort the
// paths into passes in which all 64 lanes run the same code.
//
// Why (measured, profiles/r01_v1_pmc_sq.json): with one thread per path only 15 % of the VALU lanes
// did useful work — paths end after very different bounce counts, only ~28 % of the rays pass the
// mesh gate at all, and the material branches serialise. So the wave keeps many more paths than
// lanes and, each round, picks ONE kind of work that enough slots are waiting for:
//   TERM  pass: finished paths (miss / depth 0): sky colour, attenuation fold, sample store, then a
//               fresh path from the global work counter (cam.rs:64-82); also fills empty slots
//   LAMB / METAL / DIEL pass: RayScattering::scatter of that one material, no branch divergence
//   ... each followed by the closest-sphere tests and the mesh bbox gate for the new ray
//   traversal : rays that passed a gate are "parked"; the 64 lanes traverse the BVH with a per-lane
//               stack in LDS and keep their traversal state in registers ACROSS shading passes, so a
//               lane that finishes is finalised and refilled (in batches) as soon as parked rays
//               exist and there is never a tail of a few long traversals holding a wave
// Which lane works on which path never affects the result: a path owns its RNG stream and its
// sample slot, and resolve_kernel adds the samples of a pixel in sample order.

// Region markers. Three analysis builds, none of them the product:
//   -DRBRT_MARKERS=1          asm comments at region boundaries (tools/static_cost.py counts instructions between them)
//   -DRBRT_REGION_TIMERS=1    every marker stamps s_memtime and adds the cycles since the previous marker to the region
//                             that just ended (per-wave accumulators in LDS, summed into DevCounters::diag[0..kNumRegions)
//                             at the end): where a wave's LIFETIME goes, waits included (tools/region_profile.py)
#define RBRT_REGIONS(X) X(init) X(finalise) X(census) X(refill) X(choose) X(burst_top) X(leaf_round) X(leaf_chunk) X(walk) \
    X(burst_end) X(pass_lists) X(term) X(gen) X(scatter_load) X(round_top) X(scatter_kind) X(spheres_gate) X(gate) X(park)
#define RBRT_REGION_ENUM(name) R_##name,
enum { RBRT_REGIONS(RBRT_REGION_ENUM) kNumRegions };
#ifndef RBRT_REGION_TIMERS
#define RBRT_REGION_TIMERS 0
#endif
#if RBRT_REGION_TIMERS
#define RBRT_MARK(name) rt_tick(R_##name)
#elif defined(RBRT_MARKERS)
#define RBRT_MARK(name) asm volatile("; @@" #name ::: "memory")
#else
#define RBRT_MARK(name)
#endif
#ifndef RBRT_FAST_GATE
#define RBRT_FAST_GATE 1  // mesh bbox gate through bbox_gate_fast (same decisions, no IEEE divisions on the common path)
#endif
#ifndef RBRT_PUSH_ORDER
#define RBRT_PUSH_ORDER 0  // 0: children pushed far-to-near (sorted); 1: nearest next, the rest in slot order
#endif
#ifndef RBRT_SPHERE_BOUND
#define RBRT_SPHERE_BOUND 1  // the triangle search of a ray that has hit a sphere starts at that hit's distance
#endif
#ifndef RBRT_MK_WAVES_PER_SIMD
#define RBRT_MK_WAVES_PER_SIMD 4  // register budget: 512 / 4 = 128 VGPRs per lane
#endif

// (F_TRI is the winning triangle while the closest hit is a mesh; while it is a SPHERE the word holds the bits of
// that hit's distance instead -- what scene.rs:27,37 compares the meshes' distances with. For a mesh hit the
// distance is a pure function of the ray and F_T, the same `length(o - (o + t*d))` expression that produced it,
// and is recomputed in the rare case that a later mesh has to be compared with an earlier one.)
enum { F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_S0, F_S1, F_ITEM, F_META, F_T, F_TRI, F_WORD, kFields };
constexpr uint32_t kCellDw = 128u, kTqDw = 64u;  // LDS behind the pool: 64 x u64 result cells, 64 x u32 triangle-test queue
constexpr uint32_t kHelpDw = 16u;                  // ... and 64 x u8 counts of helper lanes per owner lane (shared traversals)
constexpr unsigned long long kNoHitKey = (unsigned long long)0x49742400u << 32;  // (t = 1000000.0f, index 0): triangle.rs:398
enum : uint32_t { ST_EMPTY = 0u, ST_TRAV = 1u, ST_TERM = 2u, ST_LAMB = 3u, ST_METAL = 4u, ST_DIEL = 5u, kNumStatus = 6u,
                  ST_BUSY = 6u /* being traversed by a lane right now */ };
constexpr int kSeqWords = kMaxPathDepth / 4;

// Wave-wide vote. HIP's __ballot / __any take an int: the predicate is first materialised with v_cndmask and then
// compared again (two VALU instructions per vote, and this kernel votes several times per traversal step); the
// builtin consumes the compare's SGPR mask directly.
__device__ __forceinline__ uint64_t wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wany(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {  // set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}
__device__ __forceinline__ float lane_get(float v, int lane_byte_index) {  // v of another lane (every lane takes part)
    return __int_as_float(__builtin_amdgcn_ds_bpermute(lane_byte_index, __float_as_int(v)));
}
// meta word: depth left [0,7) | records [7,14) | object id + 1 [14,22) | mesh index [22,30)
__device__ __forceinline__ uint32_t pack_meta(uint32_t depth, uint32_t nrec, int32_t obj, uint32_t mesh) {
    return depth | (nrec << 7) | (uint32_t(obj + 1) << 14) | (mesh << 22);
}

// Scene constants staged in LDS at kernel start (a copy of the DevSphere / DevMaterial / DevMesh
// arrays as dwords), so that the shading passes do not chain dependent global loads.
constexpr uint32_t kSphDw = sizeof(DevSphere) / 4, kMatDw = sizeof(DevMaterial) / 4, kMeshDw = sizeof(DevMesh) / 4;
constexpr uint32_t kTriDw = sizeof(DevTriangle) / 4;  // v0, e0, e1, normal
static_assert(sizeof(DevSphere) == 16 && sizeof(DevMaterial) == 20 && sizeof(DevMesh) == 80, "LDS scene table layout");
enum { MD_NODES = 0, MD_TRIS = 2, MD_NORMALS = 4, MD_BBOX_LO = 6, MD_BBOX_HI = 9, MD_CENTER = 12, MD_RADIUS = 15 };
static_assert(offsetof(DevMesh, tris) == 8 && offsetof(DevMesh, normals) == 16 && offsetof(DevMesh, bbox_lo) == 24 &&
                  offsetof(DevMesh, bbox_hi) == 36 && offsetof(DevMesh, center) == 48 && offsetof(DevMesh, radius) == 60,
              "LDS scene table layout");
// path-generation parameters in LDS (behind the scene tables), as dword indices
enum { G_POS = 0, G_RIGHT = 3, G_UP = 6, G_CENTER = 9, G_MMH = 12, G_MMV, G_W, G_H, G_BATCH, G_BATCH_MAGIC, G_TILES_X,
       G_TILES_X_MAGIC, G_TILE_WORLD, G_TILE_RANK, G_N_LOCAL, G_REVERSED, G_SAMPLE_BASE, G_MAX_DEPTH, G_SEED_LO, G_SEED_HI, kGenDw };
struct SceneLds {
    const float* sph;      // [n_spheres][4]: centre xyz, radius
    const uint32_t* mat;   // [n_elem + n_meshes][5]: albedo xyz, param, kind (n_elem = spheres + BasicTriangle elements)
    const uint32_t* mesh;  // [n_meshes][20]: DevMesh as dwords
    const float* tri;      // [n_elem_tris][12]: v0, e0, e1, normal of the BasicTriangle elements (only when there are any)
    const uint32_t* elem;  // [n_elem] Scene::elements order, bit 31 = triangle (only when there are triangles: else element e = sphere e)
};
template <class T>
__device__ __forceinline__ const T* lds_ptr(const uint32_t* p) {  // a 64-bit device pointer stored as two dwords
    return reinterpret_cast<const T*>(uintptr_t(p[0]) | (uintptr_t(p[1]) << 32));
}

// First mesh with index >= m0 whose bbox gate (aabbox.rs:28-58) the ray passes, or n_meshes.
// `closest`: the distance of the ray's closest hit so far (f32::MAX: none); a mesh whose box the ray enters only
// beyond it (relaxed like the search bound of the refill) is passed over.
template <bool STATS>
__device__ __forceinline__ uint32_t next_gated_mesh(const SceneLds& sc, uint32_t n_meshes, uint32_t m0, V3 o, V3 d,
                                                    float closest, LocalCounters& lc) {
    const float beyond = RBRT_SPHERE_BOUND
                             ? closest * 1.001f + 0.001f * (1.0f + __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)),
                                                                                  __builtin_fabsf(o.z)))
                             : __builtin_inff();
    uint32_t m = m0;
    for (; m < n_meshes; ++m) {
        const float* md = reinterpret_cast<const float*>(sc.mesh + m * kMeshDw);
        if (RBRT_FAST_GATE ? bbox_gate_fast(md + MD_BBOX_LO, md + MD_BBOX_HI, o, d, beyond) : bbox_gate(md + MD_BBOX_LO, md + MD_BBOX_HI, o, d)) {
            if (STATS) ++lc.gate;
            break;
        }
    }
    return m;
}

// What has to happen next to a path whose closest hit is known.
__device__ __forceinline__ uint32_t classify(const SceneLds& sc, int32_t obj, uint32_t depth) {
    if (obj < 0 || depth == 0) return ST_TERM;  // lib.rs:54,68
    return ST_LAMB + sc.mat[uint32_t(obj) * kMatDw + 4];
}

// Work items (tile, sample, pixel) are handed out in chunks of 64 from kWorkShards global counters. A wave starts
// on the shard of its XCD (the counters do not all sit on one address: a single word saturates near 90 atomics/us,
// MI355X_MICROARCH.md "dequeue"), and moves on to the next shard when its own is exhausted, so every item is handed
// out exactly once whatever the placement is. The next chunk is requested one TERM pass ahead, so the atomic's
// latency is off the critical path.
// Shards INTERLEAVE over the item range in stripes of P.work_stripes chunks (0: contiguous eighths, the round-1
// layout): stripe b of shard k is global stripe 8b + k, so all shards sweep the image in the same order and the
// launch ENDS on the tiles api.cpp put last (the empty ones: short paths, a short drain). With contiguous eighths
// the cheap shards ran dry first and the launch ended wherever the longest paths are.
constexpr uint32_t kWorkChunk = 64;
struct WorkSource {
    uint32_t res_next = 0, res_end = 0;  // wave-uniform: the chunk being handed out (global item numbers)
    uint32_t shard = 0;                  // wave-uniform: shard to draw from next
    uint32_t pend_shard = 0;             // wave-uniform: shard of the request in flight
    uint32_t n_dry = 0;                  // wave-uniform: consecutive shards found exhausted
    unsigned long long pend_base = 0;    // lane 0: result of the request in flight
    bool pending = false;                // wave-uniform

    uint32_t n_chunks = 0;               // wave-uniform: chunks of the launch (its work items / 64)

    __device__ __forceinline__ void init(uint32_t n_items) {
        n_chunks = n_items >> 6;  // (a multiple of the chunk size: 64 pixel slots per tile)
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        shard = xcc % kWorkShards;
    }
    // (everything below is wave-uniform arithmetic on 32-bit values -- n_items < 2^32, api.cpp -- and is FORCED into
    // scalar registers with readfirstlane: left to itself the compiler kept this state in VGPRs, because the request in
    // flight lives in lane 0 only, and turned the chunk bookkeeping of every TERM pass into ~300 exec-masked vector and
    // scalar instructions with 64-bit compares)
    static __device__ __forceinline__ uint32_t uni(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
    __device__ __forceinline__ uint32_t shard_lo(uint32_t k) const {
        if (k >= kWorkShards) return n_chunks * kWorkChunk;
        return uint32_t((uint64_t(n_chunks) * k) / kWorkShards) * kWorkChunk;
    }
    // HELPER (a helper launch, api.cpp "Elastic launches"): the counters are this launch's only while the lane's last
    // RESOLVED launch is the one before it -- once this launch's own resolve has published its number the counters have
    // been handed to the next launch, and a helper wave that arrives late must not draw from them: it sees every shard dry.
    template <bool HELPER>
    __device__ __forceinline__ void prefetch(const TraceParams& P, uint32_t lane) {
        if (pending) return;
        if (lane == 0) {
            if (HELPER && __hip_atomic_load(P.helper_words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != P.helper_seq - 1u)
                pend_base = ~0ull;
            else
                pend_base = atomicAdd(P.work_counter + shard * kWorkCounterStride, (unsigned long long)kWorkChunk);
        }
        pend_shard = shard;
        pending = true;
    }
    // Blocks until a chunk is in hand: [lo, hi) global items. Returns false when every shard is exhausted.
    template <bool HELPER>
    __device__ __forceinline__ bool next_chunk(const TraceParams& P, uint32_t lane, uint32_t& lo, uint32_t& hi) {
        for (;;) {
            prefetch<HELPER>(P, lane);
            // (lane 0's result, through SGPRs. next_chunk runs in wave-uniform control flow, so the first active lane IS lane 0.)
            const uint32_t l_lo = uni(uint32_t(pend_base)), l_hi = uni(uint32_t(pend_base >> 32));
            const uint32_t ps = uni(pend_shard);
            pending = false;
            if (l_hi == 0u) {  // (a counter beyond 2^32 is far beyond every shard's end)
                if (P.work_stripes) {  // (a power of two: api.cpp)
                    const uint32_t sh = 31u - uint32_t(__builtin_clz(P.work_stripes));
                    const uint32_t l = l_lo >> 6;  // chunk number inside the shard (< 2^26)
                    const uint32_t g = (((l >> sh) * kWorkShards + ps) << sh) + (l & (P.work_stripes - 1u));  // < 8 l + 8 stripes: fits
                    if (g < n_chunks) {
                        lo = g * kWorkChunk;
                        hi = lo + kWorkChunk;
                        n_dry = 0;
                        return true;
                    }
                } else {
                    const uint32_t base = shard_lo(ps), end = shard_lo(ps + 1u);
                    if (l_lo < end - base) {
                        lo = base + l_lo;
                        hi = lo + kWorkChunk < end ? lo + kWorkChunk : end;
                        n_dry = 0;
                        return true;
                    }
                }
            }
            shard = uni((ps + 1u) % kWorkShards);  // this shard is exhausted: move on
            n_dry = uni(n_dry + 1u);
            if (n_dry >= kWorkShards) {
                lo = hi = 0;
                return false;
            }
        }
    }
};

// SHAREK: the build of the kernel that can share traversals between lanes in the drain (below). It is a separate
// build because the second copy of the traversal loop costs the first one 3 % (register allocation at the
// 128-VGPR limit): launches that are mostly bulk use the build without it.
template <int POOLN, bool STATS, bool SHAREK, bool HELPER = false>
__global__ __launch_bounds__(64, RBRT_MK_WAVES_PER_SIMD) void trace_megakernel(const TraceParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr uint32_t kPoolPad = (uint32_t(POOLN) + 63u) & ~63u;  // census loops run in groups of 64 slots
    uint32_t* const pool = lds;
    static_assert((kFields * POOLN) % 2 == 0, "the u64 cells must be 8-byte aligned");
    // Leaf rounds (below): cell[l] is the best (t, triangle index) found so far for the ray that lane l OWNS, merged
    // by whichever lanes tested its triangles; tq is the queue of (triangle, holder lane) pairs of the chunk being
    // tested; helpers[l] counts the lanes that are walking a part of lane l's tree for it (shared traversals).
    unsigned long long* const cell = reinterpret_cast<unsigned long long*>(pool + kFields * POOLN);  // [64]
    uint32_t* const tq = pool + kFields * POOLN + kCellDw;                                            // [64]
    uint32_t* const helpers = tq + kTqDw;                                         // [64] bytes, four to a word
    uint8_t* const status = reinterpret_cast<uint8_t*>(helpers + kHelpDw);       // [kPoolPad] one byte per slot
    uint8_t* const list = status + kPoolPad;                                     // [kPoolPad] slot ids (< 256)
    const uint32_t lane = threadIdx.x;
    // The launch's tiles: all of the rank's, or what the tile pass (kernels.hip tile_lists_kernel) left of them; the
    // count of the latter is known on the device only. Work items and the sample buffer are laid out over THESE tiles.
    const uint32_t n_work = P.tile_lists ? uint32_t(__builtin_amdgcn_readfirstlane(int(P.tile_lists[0]))) : P.n_local_tiles;
    const uint32_t npix = n_work * 64u;
    const uint32_t n_items = npix * P.batch;  // (< 2^32: api.cpp sizes the batches for all of the rank's tiles)
    uint32_t* const stack_base = helpers + kHelpDw + kPoolPad / 2u;                     // 2 * kPoolPad bytes of byte arrays
    uint32_t* const stack = stack_base + lane;
#if RBRT_REGION_TIMERS
    // (the accumulators sit at the very end of the workgroup's LDS: megakernel_lds_bytes adds room for them)
    // (u32: a wave spends at most a few million cycles in a region per launch; 76 bytes fit the slack of the product's
    // allocation granule, so this build keeps the product's 16 workgroups per CU)
    uint32_t* const rt_acc = lds + (megakernel_lds_dwords(POOLN, P.stack_entries, P.n_spheres, P.n_meshes, P.n_elem_tris) - uint32_t(kNumRegions));
    if (lane < uint32_t(kNumRegions)) rt_acc[lane] = 0u;
    unsigned long long rt_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_wall0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, the same clock on every CU
    unsigned long long rt_wall_workout = 0;
    uint32_t rt_dr_rounds = 0, rt_dr_passes = 0, rt_dr_steps = 0, rt_dr_live = 0, rt_dr_maxb = 0;  // this wave's drain
    uint32_t rt_cur = R_init;
    auto rt_tick = [&](uint32_t next) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) rt_acc[rt_cur] += uint32_t(now - rt_prev);
        rt_prev = now;
        rt_cur = next;
    };
#endif
    // (a helper launch's waves take the scratch slots behind those of the launch they help)
    const uint32_t wave_slot = HELPER ? blockIdx.x + P.wave_base : blockIdx.x;
    uint32_t* const gseq = P.gseq + size_t(wave_slot) * uint32_t(POOLN) * kSeqWords;
    if (HELPER) {
        // registered BEFORE the first draw from the counters: a helper wave that holds work is counted in helper_words[0] by
        // the time the launch's last work item is handed out, and the resolve waits for that count to return to zero
        uint32_t before = 0u;
        if (lane == 0) before = __hip_atomic_fetch_add(P.helper_words, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(int(before)) == -1) return;  // (never: the value is waited for, that is all)
        // Is there enough left to be worth joining? A wave that joins fills a pool of its own and, when the work runs out,
        // runs it empty alone -- half a millisecond and more at a tenth of the lane utilisation. With plenty of work left per
        // wave that is repaid many times over; joined late it only turns work the launch's own waves would have done in full
        // passes into one more drain, and the launch ends LATER (api.cpp helper_min_items has the measurements). What is left = the items no shard counter has handed out yet.
        if (P.helper_min_items != 0u) {
            unsigned long long drawn = 0ull;
            const unsigned long long per_shard = (unsigned long long)(n_items / kWorkShards) + kWorkChunk;
            for (uint32_t k = 0; k < kWorkShards; ++k) {
                const unsigned long long c = __hip_atomic_load(P.work_counter + k * kWorkCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                drawn += c < per_shard ? c : per_shard;
            }
            const unsigned long long left = drawn < n_items ? n_items - drawn : 0ull;
            if (left < (unsigned long long)(P.wave_base + gridDim.x) * P.helper_min_items) {
                if (lane == 0) __hip_atomic_fetch_sub(P.helper_words, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
#define POOL(f, s) pool[(f) * POOLN + (s)]

    for (uint32_t s = lane; s < kPoolPad; s += 64) status[s] = s < uint32_t(POOLN) ? ST_EMPTY : ST_BUSY;  // pad slots never match
    cell[lane] = kNoHitKey;
    if (lane < kHelpDw) helpers[lane] = 0u;
    // scene tables behind the stacks
    const uint32_t n_elem = P.n_spheres + P.n_elem_tris;  // Scene::elements: spheres and BasicTriangles, in the scene's order
    const uint32_t n_obj = n_elem + P.n_meshes;
    uint32_t* const sc_base = stack_base + P.stack_entries * 64u;
    {
        const uint32_t* gs = reinterpret_cast<const uint32_t*>(P.spheres);
        const uint32_t* gm = reinterpret_cast<const uint32_t*>(P.materials);
        const uint32_t* gh = reinterpret_cast<const uint32_t*>(P.meshes);
        uint32_t* dst = sc_base;
        for (uint32_t i = lane; i < P.n_spheres * kSphDw; i += 64) dst[i] = gs[i];
        dst += P.n_spheres * kSphDw;
        for (uint32_t i = lane; i < n_obj * kMatDw; i += 64) dst[i] = gm[i];
        dst += n_obj * kMatDw;
        for (uint32_t i = lane; i < P.n_meshes * kMeshDw; i += 64) dst[i] = gh[i];
        // what path generation reads (camera, work decomposition): kept in LDS rather than in ~30 SGPRs that the
        // rest of the kernel would have to spill around (the spill code is VALU: v_readlane / v_writelane)
        dst += P.n_meshes * kMeshDw;
        uint32_t* const gen_dst = dst;
        if (P.n_elem_tris != 0u) {  // behind the generation parameters: the triangle elements and the element order
            const uint32_t* gt = reinterpret_cast<const uint32_t*>(P.elem_tris);
            uint32_t* td = gen_dst + kGenDw;
            for (uint32_t i = lane; i < P.n_elem_tris * kTriDw; i += 64) td[i] = gt[i];
            td += P.n_elem_tris * kTriDw;
            for (uint32_t i = lane; i < n_elem; i += 64) td[i] = P.elems[i];
        }
        if (lane == 0) {
            float* gf = reinterpret_cast<float*>(dst);
            for (int c = 0; c < 3; ++c) {
                gf[G_POS + c] = P.cam.position[c], gf[G_RIGHT + c] = P.cam.right[c];
                gf[G_UP + c] = P.cam.up[c], gf[G_CENTER + c] = P.cam.img_center_point[c];
            }
            gf[G_MMH] = P.cam.mm_per_pix_hor, gf[G_MMV] = P.cam.mm_per_pix_vert;
            dst[G_W] = P.cam.img_width_pix, dst[G_H] = P.cam.img_height_pix;
            dst[G_BATCH] = P.batch, dst[G_BATCH_MAGIC] = P.batch_magic;
            dst[G_TILES_X] = P.tiles_x, dst[G_TILES_X_MAGIC] = P.tiles_x_magic;
            dst[G_TILE_WORLD] = P.tile_world, dst[G_TILE_RANK] = P.tile_rank;
            dst[G_N_LOCAL] = n_work, dst[G_REVERSED] = P.tiles_reversed;
            dst[G_SAMPLE_BASE] = P.sample_base, dst[G_MAX_DEPTH] = P.max_depth;
            dst[G_SEED_LO] = uint32_t(P.seed_key), dst[G_SEED_HI] = uint32_t(P.seed_key >> 32);
        }
    }
    const uint32_t* const gp = sc_base + P.n_spheres * kSphDw + n_obj * kMatDw + P.n_meshes * kMeshDw;
    const float* const gpf = reinterpret_cast<const float*>(gp);
    const SceneLds sc = {reinterpret_cast<const float*>(sc_base), sc_base + P.n_spheres * kSphDw,
                         sc_base + P.n_spheres * kSphDw + n_obj * kMatDw, reinterpret_cast<const float*>(gp + kGenDw),
                         gp + kGenDw + P.n_elem_tris * kTriDw};
    // work items are reserved from the global counter in chunks, the next chunk asynchronously
    WorkSource work;
    work.init(n_items);
    LocalCounters lc = {0, 0, 0, 0, 0};
    uint32_t n_samples_done = 0;
    uint32_t dg_pass[kNumStatus] = {0, 0, 0, 0, 0, 0}, dg_lanes[kNumStatus] = {0, 0, 0, 0, 0, 0};
    uint32_t dg_steps = 0, dg_lane_steps = 0, dg_refills = 0, dg_census = 0;
    uint32_t dg_parked_at_burst = 0, dg_shade_at_burst = 0;
    uint32_t dg_mix_free = 0, dg_mix_take8 = 0, dg_mix_pass8 = 0, dg_mix_take16 = 0, dg_mix_pass16 = 0;
    uint32_t dg_dr_rounds = 0, dg_dr_steps = 0, dg_dr_passes = 0, dg_dr_lane_steps = 0;  // after the work ran out
    unsigned long long dg_t_trav = 0, dg_t_shade = 0, dg_t0 = 0, dg_tk = 0;
    uint32_t dg_share_rounds = 0, dg_share_given = 0, dg_root_only = 0, dg_sph_tails = 0, dg_sph_pairs = 0;
    uint32_t dg_leaf_rounds = 0, dg_leaf_lanes = 0, dg_walk_rounds = 0, dg_walk_lanes = 0, dg_shade_rounds = 0;
    unsigned long long dg_rt0 = 0, dg_rt_workout = 0;
    if (STATS) {
        dg_t0 = __builtin_amdgcn_s_memtime();
        dg_rt0 = __builtin_amdgcn_s_memrealtime();
    }
    bool more_work = true;  // wave-uniform: the global work counter has not run out yet
    const float eps = P.min_dist;

    // ---- per-lane traversal state; lives in registers across shading passes ----
    // (flags are integers in VGPRs and every vote below is a direct compare of one register: a vote on a composed
    // bool costs two extra VALU instructions, v_cndmask + v_cmp, and the traversal step votes five times.
    // Invariant: a lane that is not traversing has t_cur == t_pend == kNoChild.)
    uint32_t t_active = 0;      // 1: this lane is in the middle of a traversal
    uint32_t t_has_result = 0;  // 1: this lane finished a traversal that is not finalised yet
    // slot [0,8) | mesh [8,16) | owner lane [16,22) | stack entries given away [22,32): the lane whose ray this
    // lane is walking is itself unless it is a HELPER (shared traversals, below); none of these is read per step
    uint32_t t_ids = lane << 16;
    V3 t_o = mk(0.0f, 0.0f, 0.0f), t_d = t_o;
    RayCull t_rc = {t_o, t_o, 0u, 16u, 32u, 0.0f, 0.0f};
    float t_best = 0.0f;
    uint32_t t_best_idx = 0;
    const BvhNode4* t_nodes = nullptr;
    uint32_t t_sp = 0;
    int32_t t_cur = kNoChild;   // node to visit next: >= 0 inner, < 0 leaf, kNoChild = none (stack ran empty)
    int32_t t_pend = kNoChild;  // a leaf reached earlier whose triangles have not been tested yet
    int32_t t_pend2 = kNoChild;  // a second one (only ever set while t_pend is)
    // The first P.stack_entries stack slots of a lane live in LDS, deeper ones in this wave's global scratch.
    uint32_t* const gstack = P.gstack + size_t(wave_slot) * kStackMax * 64u + lane;
    const uint32_t n_lds_stack = P.stack_entries;
    // (volatile LDS accesses: otherwise the two arms are merged into one access through a selected generic
    // pointer, i.e. a flat_store / flat_load plus a dozen address instructions on the hottest path of the kernel)
    typedef volatile __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* const stack_lds = (lds_u32*)stack;
    uint32_t dg_stack_spills = 0, dg_stack_deepest = 0;  // (counting build, per lane: pushes beyond the LDS part; the deepest stack)
    auto push = [&](int32_t v) {
        if (__builtin_expect(t_sp < n_lds_stack, 1)) {
            stack_lds[t_sp * 64u] = uint32_t(v);
        } else {
            gstack[(t_sp - n_lds_stack) * 64u] = uint32_t(v);
            if (STATS) ++dg_stack_spills;
        }
        ++t_sp;
        if (STATS) dg_stack_deepest = dg_stack_deepest > t_sp ? dg_stack_deepest : t_sp;
    };
    auto stack_at = [&](uint32_t i) -> int32_t {
        if (__builtin_expect(i < n_lds_stack, 1)) return int32_t(stack_lds[i * 64u]);
        return int32_t(gstack[(i - n_lds_stack) * 64u]);
    };
    auto pop = [&]() -> int32_t { return stack_at(--t_sp); };
    auto id_slot = [&]() { return t_ids & 255u; };
    auto id_mesh = [&]() { return (t_ids >> 8) & 255u; };
    auto id_owner = [&]() { return (t_ids >> 16) & 63u; };
    auto id_given = [&]() { return t_ids >> 22; };

    for (;;) {
        RBRT_MARK(finalise);
        // ---- finalise finished traversals in a batch (mesh.rs:245-266, scene.rs:33-41) ----
        if (wany(t_has_result != 0u)) {
            if (t_has_result) {
                const uint32_t slot = id_slot(), t_mesh = id_mesh();
                const uint32_t meta = POOL(F_META, slot);
                int32_t obj = int32_t((meta >> 14) & 255u) - 1;
                float closest = 3.40282347e+38f;  // f32::MAX (scene.rs:21)
                if (obj >= 0) {                   // dist_from_ray_orig of the closest hit so far
                    if (uint32_t(obj) < n_elem) {
                        closest = __uint_as_float(POOL(F_TRI, slot));  // an element (sphere / triangle): stored by the pass that found it
                    } else {                      // an earlier mesh of this ray: recomputed as it was computed
                        const V3 pc = t_o + __uint_as_float(POOL(F_T, slot)) * t_d;
                        closest = length(t_o - pc);
                    }
                }
                if (t_best > eps && t_best < 100000.0f && (!RBRT_SPHERE_BOUND || t_best_idx != 0xFFFFFFFFu)) {  // triangle.rs:405
                    const V3 p = t_o + t_best * t_d;
                    const float dist = length(t_o - p);
                    if (dist > P.min_dist && dist < P.max_dist) {
                        if (STATS) ++lc.mesh_hits;
                        if (dist < closest) {
                            closest = dist;
                            obj = int32_t(n_elem + t_mesh);
                            POOL(F_T, slot) = __float_as_uint(t_best);
                            POOL(F_TRI, slot) = t_best_idx;
                        }
                    }
                }
                const uint32_t m2 = next_gated_mesh<STATS>(sc, P.n_meshes, t_mesh + 1u, t_o, t_d, closest, lc);
                const uint32_t depth = meta & 127u;
                POOL(F_META, slot) = pack_meta(depth, (meta >> 7) & 127u, obj, m2 < P.n_meshes ? m2 : 0u);
                status[slot] = m2 < P.n_meshes ? ST_TRAV : classify(sc, obj, depth);
                t_has_result = 0;
            }
        }
        __syncthreads();
        RBRT_MARK(census);
        // ---- census: how many slots wait for each kind of work ---------------------------------
        uint32_t cnt[kNumStatus] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t g = 0; g < kPoolPad; g += 64) {
            const uint32_t st = status[g + lane];
#pragma unroll
            for (uint32_t k = 0; k < kNumStatus; ++k) cnt[k] += uint32_t(__popcll(wballot(st == k)));
        }
        if (STATS) ++dg_census, dg_dr_rounds += more_work ? 0u : 1u;
#if RBRT_REGION_TIMERS
        if (!more_work) {
            if (rt_dr_rounds == 0) rt_dr_live = uint32_t(POOLN) - cnt[ST_EMPTY];  // paths in hand when the work ran out
            ++rt_dr_rounds;
        }
#endif
        uint32_t n_active = uint32_t(__popcll(wballot(t_active != 0u)));

        RBRT_MARK(refill);
        // ---- idle lanes take parked rays (in batches: only when enough lanes are idle) ----
        // with plenty of parked rays the lanes are topped up sooner (y_high_water) than when few wait
        const uint32_t y_refill = cnt[ST_TRAV] >= P.y_high_min_parked ? P.y_high_water : P.y_low_water;
        if (cnt[ST_TRAV] != 0 && (n_active < y_refill || n_active + cnt[ST_TRAV] <= 64u)) {
            uint32_t ny = 0;
#pragma unroll
            for (uint32_t g = 0; g < kPoolPad; g += 64) {
                const bool m = status[g + lane] == ST_TRAV;
                const uint64_t mask = wballot(m);
                if (m) list[ny + lane_rank(mask)] = uint8_t(g + lane);
                ny += uint32_t(__popcll(mask));
            }
            __syncthreads();
            if (STATS) ++dg_refills;
            const uint64_t idle = wballot(t_active == 0u);
            if (!t_active) {
                const uint32_t k = lane_rank(idle);
                if (k < ny) {
                    const uint32_t slot = list[k];
                    status[slot] = ST_BUSY;
                    t_o = mk(__uint_as_float(POOL(F_OX, slot)), __uint_as_float(POOL(F_OY, slot)),
                             __uint_as_float(POOL(F_OZ, slot)));
                    t_d = mk(__uint_as_float(POOL(F_DX, slot)), __uint_as_float(POOL(F_DY, slot)),
                             __uint_as_float(POOL(F_DZ, slot)));
                    const uint32_t r_meta = POOL(F_META, slot);
                    const uint32_t t_mesh = (r_meta >> 22) & 255u;
                    t_ids = slot | (t_mesh << 8) | (lane << 16);
                    const uint32_t* md = sc.mesh + t_mesh * kMeshDw;
                    t_nodes = lds_ptr<BvhNode4>(md + MD_NODES);
                    t_rc = make_cull(t_o, t_d, reinterpret_cast<const float*>(md) + MD_CENTER,
                                     __uint_as_float(md[MD_RADIUS]), P.eps_frac);
                    t_best = 1000000.0f;  // triangle.rs:398
                    t_best_idx = 0;
#if RBRT_SPHERE_BOUND
                    // A ray that has hit a sphere (or an earlier mesh) already can only take a triangle that is CLOSER (scene.rs:37): the
                    // search starts at that distance instead of 1e6. The mesh's distance is length(o - (o + t d))
                    // with |d| = 1 to a few ulps, i.e. t up to rounding of the order 1e-7 (t + |o|); the bound is
                    // relaxed by 1e-3 relative and 1e-3 (1 + max |o|) absolute, so a triangle beyond it is certain
                    // to fail the exact `dist < closest` that the finalise step still applies to whatever is found.
                    // Index 0xFFFFFFFF (no triangle has it) marks "nothing found below the bound".
                    {
                        const int32_t r_obj = int32_t((r_meta >> 14) & 255u) - 1;
                        if (r_obj >= 0) {  // (a sphere: F_TRI is its distance; an earlier mesh: F_T is its t, equal to
                                           // its distance to the same few ulps)
                            const float omax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(t_o.x), __builtin_fabsf(t_o.y)),
                                                               __builtin_fabsf(t_o.z));
                            const float ref = __uint_as_float(uint32_t(r_obj) < n_elem ? POOL(F_TRI, slot) : POOL(F_T, slot));
                            const float bound = ref * 1.001f + 0.001f * (1.0f + omax);
                            if (bound < 100000.0f) t_best = bound, t_best_idx = 0xFFFFFFFFu;
                        }
                    }
#endif
                    // (a degenerate ray -- NaN / inf / zero direction -- takes no mesh hit, like the reference's ordered
                    // compares give it none: its walk does not start, and the bound below eps fails the accept test)
                    const bool r_ok = ray_is_traversable(t_o, t_d);
                    if (!r_ok) t_best = -1.0f, t_best_idx = 0xFFFFFFFFu;
                    t_sp = 0;
                    cell[lane] = ((unsigned long long)__float_as_uint(t_best) << 32) | t_best_idx;
                    t_cur = r_ok ? 0 : kNoChild;
                    t_pend = kNoChild;
                    t_pend2 = kNoChild;
                    t_active = 1;
                    if (STATS) ++dg_lanes[ST_TRAV];
                }
            }
            const uint32_t taken = uint32_t(__popcll(idle)) < ny ? uint32_t(__popcll(idle)) : ny;
            cnt[ST_TRAV] -= taken;
            n_active += taken;
        }

        RBRT_MARK(choose);
        // ---- pick the shading kind with the most waiting slots ----
        // A TERM pass also starts new paths in empty slots while work is left.
        const uint32_t n_gen_slots = more_work ? cnt[ST_EMPTY] : 0u;
        uint32_t kind = ST_TERM, best = cnt[ST_TERM] + n_gen_slots;
        if (cnt[ST_LAMB] > best) kind = ST_LAMB, best = cnt[ST_LAMB];
        if (cnt[ST_METAL] > best) kind = ST_METAL, best = cnt[ST_METAL];
        if (cnt[ST_DIEL] > best) kind = ST_DIEL, best = cnt[ST_DIEL];
        if (best == 0 && n_active == 0) break;  // nothing waits, nothing runs, no work left

        // Traverse while the lanes are well filled; shade when they are not (that is what parks new
        // rays) or when shading work has piled up to a full wave.
        // (once the work items have run out -- the drain -- there is nothing to keep lanes filled FOR: what counts
        // is how few, and how full, the remaining rounds are. drain_mode bit 1: all traversals first.)
        const bool drain = !more_work;
        const bool traverse = n_active != 0 && (best == 0 || (n_active >= P.y_low_water && best < 64u) ||
                                                (drain && (P.drain_mode & 2u)));
        if (traverse) {
            if (STATS) {
                ++dg_pass[ST_TRAV];
                if (best == 0) ++dg_pass[ST_EMPTY];  // diag[0]: traversal entered because nothing else to do
                dg_lanes[ST_EMPTY] += n_active;       // diag[6]: sum of busy lanes at entry
                dg_parked_at_burst += cnt[ST_TRAV];   // diag[55]: rays parked (no lane yet) at entry
                dg_shade_at_burst += best;            // diag[56]: slots of the fullest shading kind at entry
                dg_tk = __builtin_amdgcn_s_memtime();
            }
            // leave as soon as enough lanes are idle to make a refill / shading pass worthwhile; when no
            // other work exists, as soon as one lane has a result (it creates shading work)
            const uint32_t y_keep = cnt[ST_TRAV] >= P.y_high_min_parked ? P.y_high_water : P.y_low_water;
            // (drain_mode bit 0: in the drain a burst runs until every lane has finished)
            const uint32_t keep = (drain && (P.drain_mode & 1u)) ? 1u
                                  : (best != 0 || cnt[ST_TRAV] != 0) ? (n_active < y_keep ? n_active : y_keep)
                                                                     : n_active;
            // (traversals are shared in the drain only: before it, a parked ray uses an idle lane better, and the
            // per-step bookkeeping of sharing costs more than the lanes it fills)
            const bool share = SHAREK && P.share_idle != 0u && drain;
            // (two copies of the loop: the bulk of a frame runs the one without any of the sharing code)
            auto burst = [&](auto share_tag) {
            constexpr bool SHARE = decltype(share_tag)::value;
        RBRT_MARK(burst_top);
            do {
#if RBRT_REGION_TIMERS
                if (!more_work) ++rt_dr_steps;
#endif
                if (STATS) {
                    ++dg_steps;
                    if (!more_work) ++dg_dr_steps, dg_dr_lane_steps += uint32_t(__popcll(wballot(t_active != 0u)));
                    dg_lane_steps += uint32_t(__popcll(wballot(t_active != 0u)));
                }
                if (SHARE) {
                    // ---- shared traversal: idle lanes take over stack entries of busy ones ----
                    // A lane with entries on its stack gives the OLDEST one (the subtree nearest the root) to an idle
                    // lane, which walks it for the same ray as a helper: same origin, direction and slot, its own
                    // stack, results merged into the owner's cell by the leaf rounds, where every lane of the ray
                    // also picks up the shrunken t_best. The BVH only culls and the cell keeps the lexicographic
                    // (t, index) minimum, so who walks which subtree in which order cannot change the result. The
                    // owner finishes when its own walk has ended and its helpers have.
                    const uint64_t busy = wballot((t_active | t_has_result) != 0u);
                    const uint64_t givers = wballot(t_active != 0u && t_sp > id_given());
                    const uint32_t n_idle = 64u - uint32_t(__popcll(busy));
                    if (n_idle >= P.share_idle && givers != 0ull) {
                        const uint32_t n_givers = uint32_t(__popcll(givers)), n_pairs = n_idle < n_givers ? n_idle : n_givers;
                        int32_t give = kNoChild;
                        if (t_active != 0u && t_sp > id_given() && lane_rank(givers) < n_pairs) {
                            // (the entry is replaced by "none": when the giver pops it, everything below has been
                            // given away too, and its walk ends as if the stack had run empty)
                            const uint32_t at = id_given();
                            give = stack_at(at);
                            if (at < n_lds_stack) stack_lds[at * 64u] = uint32_t(kNoChild);
                            else gstack[(at - n_lds_stack) * 64u] = uint32_t(kNoChild);
                            t_ids += 1u << 22;
                            tq[lane_rank(givers)] = lane;
                            atomicAdd(&helpers[id_owner() >> 2], 1u << ((id_owner() & 3u) * 8u));
                        }
                        __syncthreads();
                        const uint32_t ir = lane_rank(~busy);
                        const bool take = (t_active | t_has_result) == 0u && ir < n_pairs;
                        const int sb = int((take ? tq[ir] : lane) << 2);
                        const V3 so = mk(lane_get(t_o.x, sb), lane_get(t_o.y, sb), lane_get(t_o.z, sb));
                        const V3 sd = mk(lane_get(t_d.x, sb), lane_get(t_d.y, sb), lane_get(t_d.z, sb));
                        const int s_give = __builtin_amdgcn_ds_bpermute(sb, give);
                        const uint32_t s_ids = uint32_t(__builtin_amdgcn_ds_bpermute(sb, int(t_ids)));
                        if (take) {
                            t_o = so, t_d = sd;
                            t_ids = s_ids & 0x3FFFFFu;  // the giver's slot, mesh and owner; nothing given away yet
                            const uint32_t* md = sc.mesh + id_mesh() * kMeshDw;
                            t_nodes = lds_ptr<BvhNode4>(md + MD_NODES);
                            t_rc = make_cull(t_o, t_d, reinterpret_cast<const float*>(md) + MD_CENTER,
                                             __uint_as_float(md[MD_RADIUS]), P.eps_frac);
                            const unsigned long long k = cell[id_owner()];
                            t_best = __uint_as_float(uint32_t(k >> 32));
                            t_best_idx = uint32_t(k);
                            t_sp = 0;
                            t_cur = s_give;
                            t_pend = t_pend2 = kNoChild;
                            t_active = 1;
                        }
                        __syncthreads();
                        if (STATS) ++dg_share_rounds, dg_share_given += n_pairs;
                    }
                }
                // Leaves are deferred: a lane that reaches a leaf remembers it (up to two pending leaves per lane)
                // and keeps walking; triangles are tested in LEAF ROUNDS. Testing later only delays the shrinking
                // of t_best, it cannot change the result.
                if (t_cur < 0 && t_cur != kNoChild && t_pend2 == kNoChild) {
                    if (t_pend == kNoChild) t_pend = t_cur; else t_pend2 = t_cur;
                    t_cur = t_sp != 0 ? pop() : kNoChild;
                }
                const bool can_walk = t_cur >= 0;
                const uint64_t walk_mask = wballot(t_cur >= 0), active_mask = wballot(t_active != 0u);
                // A lane is stalled when both its pending places are taken and it has reached another leaf, or when
                // its walk has ended with leaves pending. A leaf round runs when enough leaves are pending to fill
                // the wave with triangles, when enough lanes are stalled, or when nobody can walk.
                const bool pend = t_pend != kNoChild;
                const uint64_t mp = wballot(t_pend != kNoChild), mp2 = wballot(t_pend2 != kNoChild);
                const uint32_t n_pend_leaves = uint32_t(__popcll(mp)) + uint32_t(__popcll(mp2));
                const uint32_t n_stalled = uint32_t(__popcll(active_mask & ~walk_mask));
                if (STATS && walk_mask != 0ull) {
                    ++dg_walk_rounds;
                    dg_walk_lanes += uint32_t(__popcll(walk_mask));
                }
                if (n_pend_leaves >= P.leaf_leaves || (n_pend_leaves != 0 && (n_stalled >= P.leaf_round || walk_mask == 0ull))) {
        RBRT_MARK(leaf_round);
                    // ---- leaf round: the pending triangles are dealt out to ALL lanes, one triangle each ----
                    // (a leaf holds 1..4 triangles and only some lanes hold a leaf: testing them where they are
                    // pending ran at a third of the lanes.) The list is every lane's first pending leaf, then every
                    // lane's second. Position p of the concatenated triangle list belongs to the owner lane whose
                    // [prefix, prefix + count) contains it; lane j of a chunk tests position B + j for its owner's
                    // ray (fetched with ds_bpermute) and merges the result into the owner's cell with one LDS
                    // 64-bit atomic min on (t bits, reference index): t > eps > 0, so the integer order of the key
                    // IS the lexicographic (t, index) order of triangle.rs:400's strict `<` scan.
                    const uint32_t pleaf = pend ? uint32_t(~t_pend) : 0u;
                    const uint32_t pleaf2 = t_pend2 != kNoChild ? uint32_t(~t_pend2) : 0u;
                    uint32_t prefix = lane_rank(mp), total1 = uint32_t(__popcll(mp));
                    uint32_t prefix2 = lane_rank(mp2), total2 = uint32_t(__popcll(mp2));
#pragma unroll
                    for (int b = 0; b < kLeafBits; ++b) {
                        const uint64_t m1 = wballot(((pleaf >> b) & 1u) != 0u), m2 = wballot(((pleaf2 >> b) & 1u) != 0u);
                        prefix += lane_rank(m1) << b;
                        total1 += uint32_t(__popcll(m1)) << b;
                        prefix2 += lane_rank(m2) << b;
                        total2 += uint32_t(__popcll(m2)) << b;
                    }
                    prefix2 += total1;
                    const uint32_t n_pend_tris = total1 + total2;
                    const uint32_t n_mine = pend ? (pleaf & uint32_t(kLeafMax - 1)) + 1u : 0u, first = pleaf >> kLeafBits;
                    const uint32_t n_mine2 = t_pend2 != kNoChild ? (pleaf2 & uint32_t(kLeafMax - 1)) + 1u : 0u, first2 = pleaf2 >> kLeafBits;
                    if (STATS) {
                        ++dg_leaf_rounds;
                        dg_leaf_lanes += n_pend_tris;
                    }
        RBRT_MARK(leaf_chunk);
                    for (uint32_t B = 0; B < n_pend_tris; B += 64u) {
#pragma unroll
                        for (uint32_t i = 0; i < uint32_t(kLeafMax); ++i) {
                            const uint32_t pos = prefix + i - B, pos2 = prefix2 + i - B;
                            if (i < n_mine && pos < 64u) tq[pos] = ((first + i) << 6) | lane;
                            if (i < n_mine2 && pos2 < 64u) tq[pos2] = ((first2 + i) << 6) | lane;
                        }
                        __syncthreads();
                        const bool valid = B + lane < n_pend_tris;
                        const uint32_t e = valid ? tq[lane] : lane;
                        const int ob = int((e & 63u) << 2);  // the lane that holds the leaf; ds_bpermute takes a byte index
                        const V3 ro = mk(lane_get(t_o.x, ob), lane_get(t_o.y, ob), lane_get(t_o.z, ob));
                        const V3 rd = mk(lane_get(t_d.x, ob), lane_get(t_d.y, ob), lane_get(t_d.z, ob));
                        const uint32_t ray_owner = SHARE ? (uint32_t(__builtin_amdgcn_ds_bpermute(ob, int(t_ids))) >> 16) & 63u : e & 63u;
                        const auto* tp = RBRT_AS1(f32x4, P.tris + (valid ? e >> 6 : 0u));
                        const f32x4 ta = tp[0], tb = tp[1];
                        const f32x2 tc = *RBRT_AS1(f32x2, tp + 2);  // the record's last 8 bytes are padding: not fetched
                        float tt;
                        const bool hit = tri_test(mk(ta.x, ta.y, ta.z), mk(ta.w, tb.x, tb.y), mk(tb.z, tb.w, tc.x), ro, rd, eps,
                                                  P.eps_frac, tt);
                        if (STATS && valid) ++lc.tris;
                        if (valid && hit)
                            atomicMin(&cell[ray_owner], ((unsigned long long)__float_as_uint(tt) << 32) | __float_as_uint(tc.y));
                        __syncthreads();
                    }
                    if (pend) {  // (the cell only ever shrinks, and holds everything this lane found before)
                        const unsigned long long k = cell[SHARE ? id_owner() : lane];
                        t_best = __uint_as_float(uint32_t(k >> 32));
                        t_best_idx = uint32_t(k);
                        t_pend = kNoChild;
                        t_pend2 = kNoChild;
                    }
                }
        RBRT_MARK(walk);
#ifdef RBRT_DUMMY_WALK  // calibration experiment: N extra full-rate VALU instructions per traversal step (the result is unused)
                {
                    float dummy = __uint_as_float(t_sp);
#pragma unroll
                    for (int i = 0; i < RBRT_DUMMY_WALK; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dummy));
                    asm volatile("" ::"v"(dummy));
                }
#endif
                if (can_walk) {
                    uint32_t k[4];
#if RBRT_PUSH_ORDER == 0 && RBRT_PAIR_SORT
                    int32_t sl[4];
                    node4_visit_sorted(t_nodes + t_cur, t_rc, eps, t_best, k, sl);
                    if (STATS) ++lc.nodes;
                    if (STATS && t_cur == 0 && k[0] == kMissKey) ++dg_root_only;  // a traversal that ends at the root
                    if (k[0] != kMissKey) {  // farthest first, so that the nearest is popped first
                        if (k[3] != kMissKey) push(sl[3]);
                        if (k[2] != kMissKey) push(sl[2]);
                        if (k[1] != kMissKey) push(sl[1]);
                        t_cur = sl[0];
                    } else {
                        t_cur = t_sp != 0 ? pop() : kNoChild;
                    }
#else
                    f32x4 links;
                    node4_visit<RBRT_PUSH_ORDER == 0>(t_nodes + t_cur, t_rc, eps, t_best, k, links);
                    if (STATS) ++lc.nodes;
                    if (STATS && t_cur == 0 && k[0] == kMissKey) ++dg_root_only;  // a traversal that ends at the root
#if RBRT_PUSH_ORDER == 0
                    if (k[0] != kMissKey) {  // farthest first, so that the nearest is popped first
                        if (k[3] != kMissKey) push(link_of(links, k[3]));
                        if (k[2] != kMissKey) push(link_of(links, k[2]));
                        if (k[1] != kMissKey) push(link_of(links, k[1]));
                        t_cur = link_of(links, k[0]);
                    } else {
                        t_cur = t_sp != 0 ? pop() : kNoChild;
                    }
#else
                    // the nearest child next, the other hit children onto the stack in slot order (no sort, no
                    // per-entry link selection); keys are distinct (the slot is in their low bits)
                    const uint32_t kmin = min(min(k[0], k[1]), min(k[2], k[3]));
                    if (kmin != kMissKey) {
                        if (k[0] != kMissKey && k[0] != kmin) push(__float_as_int(links.x));
                        if (k[1] != kMissKey && k[1] != kmin) push(__float_as_int(links.y));
                        if (k[2] != kMissKey && k[2] != kmin) push(__float_as_int(links.z));
                        if (k[3] != kMissKey && k[3] != kmin) push(__float_as_int(links.w));
                        t_cur = link_of(links, kmin);
                    } else {
                        t_cur = t_sp != 0 ? pop() : kNoChild;
                    }
#endif
#endif
                }
                if (t_active && t_cur == kNoChild && t_pend == kNoChild) {
                    if (!SHARE) {  // (no helpers exist outside shared bursts: the registers hold the result)
                        t_active = 0;
                        t_has_result = 1;
                    } else if (id_owner() != lane) {  // a helper: what it found is in the owner's cell already
                        atomicSub(&helpers[id_owner() >> 2], 1u << ((id_owner() & 3u) * 8u));
                        t_ids = lane << 16;
                        t_active = 0;
                    } else if (((helpers[lane >> 2] >> ((lane & 3u) * 8u)) & 255u) == 0u) {  // (else: wait for the helpers)
                        const unsigned long long k = cell[lane];  // with what the helpers found
                        t_best = __uint_as_float(uint32_t(k >> 32));
                        t_best_idx = uint32_t(k);
                        t_active = 0;
                        t_has_result = 1;
                    }
                }
        RBRT_MARK(burst_end);
            } while (uint32_t(__popcll(wballot(t_active != 0u))) >= keep);
            };
            if (share)
                burst(std::true_type{});
            else
                burst(std::false_type{});
            if (STATS) dg_t_trav += __builtin_amdgcn_s_memtime() - dg_tk;
            continue;
        }

        RBRT_MARK(pass_lists);
        // ============================ shading pass of one kind ===============================
#if RBRT_REGION_TIMERS
        if (!more_work) ++rt_dr_passes;
#endif
        if (STATS) {
            ++dg_pass[kind];
            if (!more_work) ++dg_dr_passes;
            dg_tk = __builtin_amdgcn_s_memtime();
        }
        uint32_t c0 = 0, c1 = 0;
#pragma unroll
        for (uint32_t g = 0; g < kPoolPad; g += 64) {
            const bool m = status[g + lane] == kind;
            const uint64_t mask = wballot(m);
            if (m) list[c0 + lane_rank(mask)] = uint8_t(g + lane);
            c0 += uint32_t(__popcll(mask));
        }
        const uint32_t n_main = c0 < 64u ? c0 : 64u;
        if (kind == ST_TERM && n_gen_slots != 0 && n_main < 64u) {
#pragma unroll
            for (uint32_t g = 0; g < kPoolPad; g += 64) {
                const bool m = status[g + lane] == ST_EMPTY;
                const uint64_t mask = wballot(m);
                if (m) list[c0 + c1 + lane_rank(mask)] = uint8_t(g + lane);  // c0 + c1 <= POOLN
                c1 += uint32_t(__popcll(mask));
            }
        }
        const uint32_t n_gen = c1 < 64u - n_main ? c1 : 64u - n_main;
        __syncthreads();
        if (STATS) {  // what a pass of mixed kinds would have picked up: the fullest OTHER scatter kind, into the free lanes
            const uint32_t free_lanes = 64u - n_main - n_gen;
            uint32_t other = 0;
            for (uint32_t k2 = ST_LAMB; k2 <= ST_DIEL; ++k2)
                if (k2 != kind && cnt[k2] > other) other = cnt[k2];
            const uint32_t take = other < free_lanes ? other : free_lanes;
            dg_mix_free += free_lanes;
            if (take >= 8u) dg_mix_take8 += take, ++dg_mix_pass8;
            if (take >= 16u) dg_mix_take16 += take, ++dg_mix_pass16;
        }
        const bool is_main = lane < n_main;
        const bool is_gen = !is_main && lane < n_main + n_gen;
        if (STATS) dg_lanes[kind] += n_main + n_gen;
        const uint32_t slot = is_main ? list[lane] : (is_gen ? list[c0 + (lane - n_main)] : 0u);

        V3 o = mk(0.0f, 0.0f, 0.0f), d = o;
        Rng rng = {0u, 0u};
        uint32_t item = 0, depth = 0, nrec = 0, word = 0;
        bool have_ray = false;

        RBRT_MARK(term);
        if (kind == ST_TERM) {
            bool need_new = is_gen;
            if (is_main) {
                const uint32_t meta = POOL(F_META, slot);
                nrec = (meta >> 7) & 127u;
                const int32_t obj = int32_t((meta >> 14) & 255u) - 1;
                V3 color = mk(0.0f, 0.0f, 0.0f);  // hit with depth 0 or a failed scatter: lib.rs:63-66
                if (obj < 0) color = background(__uint_as_float(POOL(F_DY, slot)), P.bg);  // lib.rs:68-71
                if (nrec != 0) {  // lib.rs:62: attenuation * colorize(...), innermost bounce first
                    word = POOL(F_WORD, slot);
                    for (uint32_t k = nrec; k-- > 0;) {
                        const uint32_t w = (k >> 2) == (nrec >> 2) ? word : gseq[size_t(slot) * kSeqWords + (k >> 2)];
                        const uint32_t ob = (w >> (8u * (k & 3u))) & 0xFFu;
                        color = mk(reinterpret_cast<const float*>(sc.mat + ob * kMatDw)) * color;
                    }
                }
                item = POOL(F_ITEM, slot);  // index of this path's sample in the sample buffer
#if RBRT_REGION_TIMERS
                if (!more_work) {  // the longest path this wave finishes in its drain
                    const uint32_t bounces = P.max_depth - (meta & 127u);
                    uint32_t mb = bounces;
                    for (int sh = 32; sh >= 1; sh >>= 1) mb = max(mb, uint32_t(__shfl_xor(int(mb), sh, 64)));
                    rt_dr_maxb = max(rt_dr_maxb, uint32_t(__builtin_amdgcn_readfirstlane(int(mb))));
                }
#endif
                if (STATS) {  // path-length histogram; bounces = scatter events, recorded or not
                    const uint32_t bounces = P.max_depth - (meta & 127u);
                    atomicAdd(&P.counters->diag[32 + (31 - __clz(int(bounces + 1u)))], 1ull);
                    if (bounces >= 16u) {
                        atomicAdd(&P.counters->diag[48], (unsigned long long)bounces);
                        atomicAdd(&P.counters->diag[47], (unsigned long long)(bounces - nrec));  // dielectric ones
                        for (uint32_t k = 0; k < nrec; ++k) {
                            const uint32_t w = (k >> 2) == (nrec >> 2) ? word : gseq[size_t(slot) * kSeqWords + (k >> 2)];
                            atomicAdd(&P.counters->diag[40 + (((w >> (8u * (k & 3u))) & 0xFFu) % 7u)], 1ull);
                        }
                    }
                }
                if (item < n_items) {
                    float* out = P.sample_buf + size_t(item) * 3u;
                    out[0] = color.x;
                    out[1] = color.y;
                    out[2] = color.z;
                } else {  // corrupt path state: never store out of bounds; the render call fails loudly (api.cpp)
                    atomicAdd(&P.counters->diag[57], 1ull);
                }
                if (STATS) ++n_samples_done;
                need_new = true;
            }
        RBRT_MARK(gen);
            // ---- new paths (cam.rs:64-82); work items come from the sharded global counters ----
            const uint64_t want = wballot(need_new && more_work);
            if (want) {
                const uint32_t n_want = uint32_t(__popcll(want));
                const uint32_t avail = work.res_end - work.res_next;
                uint32_t new_lo = 0, new_hi = 0;
                if (avail < n_want) {
                    more_work = work.template next_chunk<HELPER>(P, lane, new_lo, new_hi);
                    // drain_mode bit 3: a wave whose launch has run out of work items issues ahead of the bulk waves of
                    // other launches on its SIMD: the drain is a chain of dependent rounds, the bulk fills the gaps
                    if (!more_work && (P.drain_mode & 8u)) __builtin_amdgcn_s_setprio(2);
                    if (STATS && !more_work) dg_rt_workout = __builtin_amdgcn_s_memrealtime();
#if RBRT_REGION_TIMERS
                    if (!more_work) rt_wall_workout = __builtin_amdgcn_s_memrealtime();
#endif
                }
                if (need_new) {
                    const uint32_t rk = lane_rank(want);
                    const uint32_t it = rk < avail ? work.res_next + rk : new_lo + (rk - avail);
                    if (rk < avail || it < new_hi) {
                        const uint32_t pp = it & 63u;
                        const uint32_t ts = it >> 6;
                        const uint32_t batch = gp[G_BATCH];
                        const uint32_t tpos = div_magic(ts, batch, gp[G_BATCH_MAGIC]);
                        const uint32_t s = ts - tpos * batch;
                        const uint32_t widx = gp[G_REVERSED] ? gp[G_N_LOCAL] - 1u - tpos : tpos;  // row-major, either way
                        item = s * npix + widx * 64u + pp;                       // < 2^32: the host sizes batches so
                        // the launch's tiles: all of the rank's, or those the tile pass left (TraceParams::tile_lists)
                        const uint32_t tile_local = P.tile_lists ? P.tile_lists[kTileListHeader + widx] : widx;
                        const uint32_t tile = tile_local * gp[G_TILE_WORLD] + gp[G_TILE_RANK];
                        const uint32_t tiles_x = gp[G_TILES_X];
                        const uint32_t ty = div_magic(tile, tiles_x, gp[G_TILES_X_MAGIC]);
                        uint32_t tx = tile - ty * tiles_x + RBRT_TILE_SKEW * ty;  // (rbrt_hip.h "How tiles are dealt to ranks")
                        tx -= div_magic(tx, tiles_x, gp[G_TILES_X_MAGIC]) * tiles_x;
                        const uint32_t row = ty * RBRT_TILE + (pp >> 3), col = tx * RBRT_TILE + (pp & 7u);
                        const uint32_t img_w = gp[G_W], img_h = gp[G_H];
                        if (row < img_h && col < img_w) {
                            rng.init((uint64_t(gp[G_SEED_HI]) << 32) | gp[G_SEED_LO], row * img_w + col, gp[G_SAMPLE_BASE] + s);
                            o = mk(gpf + G_POS);
                            d = camera_ray_direction(o, mk(gpf + G_CENTER), mk(gpf + G_RIGHT), mk(gpf + G_UP), gpf[G_MMH], gpf[G_MMV], img_w,
                                                     img_h, row, col, rng);
                            depth = gp[G_MAX_DEPTH];
                            nrec = 0;
                            word = 0;
                            have_ray = true;
                        }
                    }
                }
                if (avail < n_want) {
                    work.res_next = WorkSource::uni(new_lo + (n_want - avail) < new_hi ? new_lo + (n_want - avail) : new_hi);
                    work.res_end = WorkSource::uni(new_hi);
                } else {
                    work.res_next = WorkSource::uni(work.res_next + n_want);
                }
                // reserve the next chunk now; its result is not needed before a later TERM pass
                if (more_work && work.res_end - work.res_next < 64u) work.template prefetch<HELPER>(P, lane);
            }
        }
        RBRT_MARK(scatter_load);
        // ---- RayScattering::scatter, then the closest sphere + mesh gate for the new ray ----
        // In the first round every lane shades the pass's kind (wave-uniform branches). A lane whose new
        // ray passes no mesh gate already knows its next hit; when that needs shading again it may stay
        // in registers for further rounds (any kind, per-lane branches) instead of waiting in the pool
        // for a later pass: chains of sphere-only bounces (a ray caught between a sphere and the ground
        // makes 30-50 of them) then cost a few us per bounce instead of one scheduling round each.
        uint32_t lk = kind;
        bool scat = kind != ST_TERM && is_main;
        int32_t s_obj = -1;
        float s_ht = 0.0f;
        uint32_t s_tri = 0;
        if (scat) {
            o = mk(__uint_as_float(POOL(F_OX, slot)), __uint_as_float(POOL(F_OY, slot)),
                   __uint_as_float(POOL(F_OZ, slot)));
            d = mk(__uint_as_float(POOL(F_DX, slot)), __uint_as_float(POOL(F_DY, slot)),
                   __uint_as_float(POOL(F_DZ, slot)));
            rng.s0 = POOL(F_S0, slot);
            rng.s1 = POOL(F_S1, slot);
            item = POOL(F_ITEM, slot);
            word = POOL(F_WORD, slot);
            const uint32_t meta = POOL(F_META, slot);
            depth = meta & 127u;
            nrec = (meta >> 7) & 127u;
            s_obj = int32_t((meta >> 14) & 255u) - 1;
            s_ht = __uint_as_float(POOL(F_T, slot));
            s_tri = POOL(F_TRI, slot);
        }
        RBRT_MARK(round_top);
        for (uint32_t round = 1;; ++round) {
            if (scat) {
                const V3 p = o + s_ht * d;  // same expression as inside the intersection routines
                V3 n;
                if (uint32_t(s_obj) < n_elem) {
                    const uint32_t desc = P.n_elem_tris != 0u ? sc.elem[uint32_t(s_obj)] : uint32_t(s_obj);
                    if (desc >> 31) n = mk(sc.tri + (desc & 0x7FFFFFFFu) * kTriDw + 9);  // triangle.rs:432: the stored normal
                    else n = p - mk(sc.sph + desc * kSphDw);                            // sphere.rs:56, unnormalised
                } else {
                    const Normal4 nn =
                        lds_ptr<Normal4>(sc.mesh + (uint32_t(s_obj) - n_elem) * kMeshDw + MD_NORMALS)[s_tri];
                    n = mk(nn.x, nn.y, nn.z);  // mesh.rs:253-257
                }
                DevMaterial m;
                {
                    const uint32_t* mp = sc.mat + uint32_t(s_obj) * kMatDw;
                    m.albedo[0] = __uint_as_float(mp[0]), m.albedo[1] = __uint_as_float(mp[1]);
                    m.albedo[2] = __uint_as_float(mp[2]), m.param = __uint_as_float(mp[3]);
                    m.kind = int32_t(mp[4]);
                }
                V3 nd;
                bool ok;
                RBRT_MARK(scatter_kind);
                if (lk == ST_LAMB) {  // lambertian.rs:11-24
                    const V3 target = (p + normalize(n)) + random_point_in_unit_sphere(rng);
                    nd = normalize(target - p);
                    ok = true;
                } else if (lk == ST_METAL) {  // metal.rs:12-25
                    const V3 target = reflect(d, n);
                    nd = normalize(target + m.param * random_point_in_unit_sphere(rng));
                    ok = dot(nd, n) > 0.0f;
                } else {  // dielectric.rs:11-59
                    ok = scatter(m, d, p, n, rng, nd);
                }
                if (ok) {
                    if (lk != ST_DIEL) {  // attenuation (1,1,1) is an exact identity, not recorded
                        word |= uint32_t(s_obj) << (8u * (nrec & 3u));
                        if ((nrec & 3u) == 3u) {
                            gseq[size_t(slot) * kSeqWords + (nrec >> 2)] = word;
                            word = 0;
                        }
                        ++nrec;
                    }
                    o = p;
                    d = nd;
                    depth -= 1;
                    have_ray = true;
                } else {
                    // metal.rs:25 returned false: the path is black (lib.rs:63-66). Park it for a TERM
                    // pass with depth 0 so that the fold/store/regenerate code lives in one place.
                    // (a path started in this very pass has nothing of its own in the pool yet: F_ITEM too)
                    POOL(F_META, slot) = pack_meta(0u, nrec, s_obj, 0u);
                    POOL(F_WORD, slot) = word;
                    POOL(F_ITEM, slot) = item;
                    status[slot] = ST_TERM;
                }
            }
            scat = false;
            // closest sphere + mesh gate for the new ray (scene.rs:19-43 up to the meshes)
            uint32_t gated = 0, next = ST_TERM;
            float closest = 3.40282347e+38f;
        RBRT_MARK(spheres_gate);
            if (have_ray) {
                if (STATS) ++lc.rays;
                s_obj = -1;
                s_ht = 0.0f;
                // Scene::elements in their order (scene.rs:23-31). Two copies of the loop: a scene without BasicTriangles
                // (every scene the reference's YAML can describe) runs the one that knows only spheres -- with the
                // element-kind test inside a single loop the sphere-only frame was 2 % slower.
                auto test_sphere = [&](uint32_t e, const float* sp) {
                    float t, dist;
                    if (STATS) {  // how often a wave runs the expensive part of sphere_hit, and for how many lanes
                        const V3 l = o - mk(sp);
                        const float bq = dot(d * 2.0f, l);
                        const float sol = bq * bq - 4.0f * dot(d, d) * (dot(l, l) - sp[3] * sp[3]);
                        const uint64_t m = wballot(sol >= 0.0f);
                        if (m != 0ull) ++dg_sph_tails, dg_sph_pairs += uint32_t(__popcll(m));
                    }
                    if (sphere_hit(mk(sp), sp[3], o, d, P.min_dist, P.max_dist, t, dist, P.counters)) {
                        if (dist < closest) {
                            closest = dist;
                            s_ht = t;
                            s_obj = int32_t(e);
                        }
                    }
                };
                if (P.n_elem_tris == 0u) {
                    for (uint32_t i = 0; i < P.n_spheres; ++i) test_sphere(i, sc.sph + i * kSphDw);
                } else {
                    for (uint32_t e = 0; e < n_elem; ++e) {
                        const uint32_t desc = sc.elem[e];  // (wave-uniform: which kind of element this is)
                        if (desc >> 31) {
                            float t, dist;
                            if (basic_triangle_hit(sc.tri + (desc & 0x7FFFFFFFu) * kTriDw, o, d, P.min_dist, P.max_dist, t, dist)) {
                                if (dist < closest) {
                                    closest = dist;
                                    s_ht = t;
                                    s_obj = int32_t(e);
                                }
                            }
                        } else {
                            test_sphere(e, sc.sph + desc * kSphDw);
                        }
                    }
                }
                RBRT_MARK(gate);
                gated = next_gated_mesh<STATS>(sc, P.n_meshes, 0, o, d, closest, lc);
                next = gated < P.n_meshes ? uint32_t(ST_TRAV) : classify(sc, s_obj, depth);
            }
            // stay in registers? only sphere hits that need shading, while enough lanes do (or the wave
            // has no other work left), and for a bounded number of rounds
            RBRT_MARK(park);
            const bool cand = have_ray && next >= ST_LAMB;
            const uint32_t n_cand = uint32_t(__popcll(wballot(cand)));
            const bool go = n_cand != 0 && round < kMaxShadeRounds &&
                            (more_work ? (round < P.shade_rounds && n_cand >= P.shade_cont_min) : true);
            if (have_ray && !(cand && go)) {
                POOL(F_OX, slot) = __float_as_uint(o.x);
                POOL(F_OY, slot) = __float_as_uint(o.y);
                POOL(F_OZ, slot) = __float_as_uint(o.z);
                POOL(F_DX, slot) = __float_as_uint(d.x);
                POOL(F_DY, slot) = __float_as_uint(d.y);
                POOL(F_DZ, slot) = __float_as_uint(d.z);
                POOL(F_S0, slot) = rng.s0;
                POOL(F_S1, slot) = rng.s1;
                POOL(F_ITEM, slot) = item;
                POOL(F_WORD, slot) = word;
                POOL(F_T, slot) = __float_as_uint(s_ht);
                POOL(F_TRI, slot) = __float_as_uint(closest);  // (meaningful while the closest hit is a sphere)
                POOL(F_META, slot) = pack_meta(depth, nrec, s_obj, gated < P.n_meshes ? gated : 0u);
                status[slot] = uint8_t(next);
            } else if (round == 1 && kind == ST_TERM && !have_ray && (is_main || is_gen)) {
                status[slot] = ST_EMPTY;  // no work item left (or a pixel outside a ragged image edge)
            }
            if (!go) break;
            scat = cand;
            have_ray = false;
            lk = next;
            s_tri = 0;
            if (STATS) dg_lanes[kind] += n_cand, ++dg_shade_rounds;
        }
        if (STATS) dg_t_shade += __builtin_amdgcn_s_memtime() - dg_tk;
    }
#undef POOL
    if (HELPER) {  // this wave's samples are written and visible to the device before it stops counting as a holder of work
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_fetch_sub(P.helper_words, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
#if RBRT_REGION_TIMERS
    rt_tick(R_init);
    if (lane < uint32_t(kNumRegions)) atomicAdd(&P.counters->diag[lane], (unsigned long long)rt_acc[lane]);
    if (lane == 0) {  // the launch's phases on the wall clock: ramp-up, bulk, drain
        const unsigned long long rt_wall1 = __builtin_amdgcn_s_memrealtime();
        if (rt_wall_workout == 0) rt_wall_workout = rt_wall1;
        atomicAdd(&P.counters->diag[kNumRegions], 1ull);                            // waves
        atomicMin(&P.counters->diag[kNumRegions + 1], rt_wall0);                     // first wave start (slot preset to ~0 by the host tool)
        atomicMax(&P.counters->diag[kNumRegions + 2], rt_wall0);                     // last wave start
        atomicMin(&P.counters->diag[kNumRegions + 3], rt_wall_workout);              // first wave out of work items
        atomicMax(&P.counters->diag[kNumRegions + 4], rt_wall_workout);              // last wave out of work items
        atomicMax(&P.counters->diag[kNumRegions + 5], rt_wall1);                     // last wave end
        atomicAdd(&P.counters->diag[kNumRegions + 6], rt_wall1 - rt_wall_workout);   // summed drain time of the waves
        atomicAdd(&P.counters->diag[kNumRegions + 7], rt_wall1 - rt_wall0);          // summed lifetime of the waves
        // when did this wave START, in 100-us buckets after the epoch the host tool put in diag[kNumRegions + 8] (the end of
        // the previous launch): waves that start late were not resident from the beginning (the grid exceeds what fits)
        const unsigned long long epoch = P.counters->diag[kNumRegions + 8];
        const unsigned long long b = rt_wall0 > epoch ? (rt_wall0 - epoch) / 10000ull : 0ull;
        atomicAdd(&P.counters->diag[32 + (b < 31ull ? b : 31ull)], 1ull);
        // the wave with the longest drain: [drain in 10-ns ticks : 24 | rounds : 10 | passes : 10 | steps : 12 | paths in hand : 8]
        // and, keyed the same way, the longest path it finished there
        const unsigned long long ticks = (rt_wall1 - rt_wall_workout) & 0xFFFFFFull;
        atomicMax(&P.counters->diag[kNumRegions + 9], (ticks << 40) | ((unsigned long long)(rt_dr_rounds & 1023u) << 30) |
                                                          ((unsigned long long)(rt_dr_passes & 1023u) << 20) |
                                                          ((unsigned long long)(rt_dr_steps & 4095u) << 8) | (rt_dr_live & 255u));
        atomicMax(&P.counters->diag[kNumRegions + 10], (ticks << 40) | rt_dr_maxb);
    }
#endif
    if (STATS) {
        atomicAdd(&P.counters->rays, (unsigned long long)lc.rays);
        atomicAdd(&P.counters->mesh_gate_pass, (unsigned long long)lc.gate);
        atomicAdd(&P.counters->nodes_visited, (unsigned long long)lc.nodes);
        atomicAdd(&P.counters->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&P.counters->mesh_hits, (unsigned long long)lc.mesh_hits);
        atomicAdd(&P.counters->samples, (unsigned long long)n_samples_done);
        atomicAdd(&P.counters->diag[61], (unsigned long long)dg_root_only);  // (per lane, like the counters above)
        if (dg_stack_spills != 0u) atomicAdd(&P.counters->diag[30], (unsigned long long)dg_stack_spills);
        atomicMax(&P.counters->diag[31], (unsigned long long)dg_stack_deepest);
        if (lane == 0) {
            for (uint32_t k = 0; k < kNumStatus; ++k) {
                atomicAdd(&P.counters->diag[k], (unsigned long long)dg_pass[k]);
                atomicAdd(&P.counters->diag[6 + k], (unsigned long long)dg_lanes[k]);
            }
            {   // the wave with the longest drain phase: [us:20 | rounds:12 | steps:16 | passes:16]
                const unsigned long long rt_now = __builtin_amdgcn_s_memrealtime();
                const unsigned long long us = dg_rt_workout ? (rt_now - dg_rt_workout) / 100ull : 0ull;
                const unsigned long long packed = (us << 44) | ((unsigned long long)(dg_dr_rounds & 0xFFFu) << 32) |
                                                  ((unsigned long long)(dg_dr_steps & 0xFFFFu) << 16) | (dg_dr_passes & 0xFFFFu);
                atomicMax(&P.counters->diag[50], packed);
                atomicAdd(&P.counters->diag[51], (unsigned long long)dg_dr_rounds);
                atomicAdd(&P.counters->diag[52], (unsigned long long)dg_dr_steps);
                atomicAdd(&P.counters->diag[53], (unsigned long long)dg_dr_passes);
                atomicAdd(&P.counters->diag[54], (unsigned long long)dg_dr_lane_steps);
            }
            atomicAdd(&P.counters->diag[55], (unsigned long long)dg_parked_at_burst);
            atomicAdd(&P.counters->diag[23], (unsigned long long)dg_mix_free);
            atomicAdd(&P.counters->diag[49], (((unsigned long long)dg_mix_pass8) << 32) | dg_mix_take8);
            atomicAdd(&P.counters->diag[58], (((unsigned long long)dg_mix_pass16) << 32) | dg_mix_take16);
            atomicAdd(&P.counters->diag[56], (unsigned long long)dg_shade_at_burst);
            atomicAdd(&P.counters->diag[12], (unsigned long long)dg_steps);
            atomicAdd(&P.counters->diag[13], (unsigned long long)dg_lane_steps);
            atomicAdd(&P.counters->diag[14], (unsigned long long)dg_refills);
            atomicAdd(&P.counters->diag[15], (unsigned long long)dg_census);
            atomicAdd(&P.counters->diag[16], dg_t_trav);
            atomicAdd(&P.counters->diag[17], dg_t_shade);
            atomicAdd(&P.counters->diag[18], (unsigned long long)(__builtin_amdgcn_s_memtime() - dg_t0));
            atomicAdd(&P.counters->diag[59], (unsigned long long)dg_share_given);
            atomicAdd(&P.counters->diag[60], (unsigned long long)dg_share_rounds);
            atomicAdd(&P.counters->diag[62], (unsigned long long)dg_sph_tails);
            atomicAdd(&P.counters->diag[63], (unsigned long long)dg_sph_pairs);
            atomicAdd(&P.counters->diag[19], (unsigned long long)dg_leaf_rounds);
            atomicAdd(&P.counters->diag[20], (unsigned long long)dg_leaf_lanes);
            atomicAdd(&P.counters->diag[29], (unsigned long long)dg_shade_rounds);
            atomicAdd(&P.counters->diag[21], (unsigned long long)dg_walk_rounds);
            atomicAdd(&P.counters->diag[22], (unsigned long long)dg_walk_lanes);
            const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
            atomicMin(&P.counters->diag[24], dg_rt0);
            atomicMax(&P.counters->diag[25], dg_rt_workout);
            atomicMax(&P.counters->diag[26], rt1);
            atomicAdd(&P.counters->diag[27], rt1 - dg_rt0);
            atomicMin(&P.counters->diag[28], dg_rt_workout ? dg_rt_workout : rt1);
        }
    }
}
