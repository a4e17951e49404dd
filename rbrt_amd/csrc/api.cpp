// api.cpp — implementation of the C ABI in include/rbrt_hip.h on top of the gfx950 kernels.
// Host C++ only (compiled by hipcc for the HIP runtime headers); no torch, no Python.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bvh.h"
#include "bvh_device.h"
#include "device_types.h"
#include "../../include/rbrt_hip_debug.h"

namespace rbrt {
hipError_t launch_trace_megakernel(const TraceParams& P, uint32_t n_waves, uint32_t pool, bool stats, bool share,
                                   hipStream_t stream);
hipError_t launch_trace_helper(const TraceParams& P, uint32_t n_waves, uint32_t pool, bool share, hipStream_t stream);
size_t megakernel_gseq_bytes(uint32_t n_waves, uint32_t pool);
size_t megakernel_gstack_bytes(uint32_t n_waves);
size_t megakernel_lds_bytes(uint32_t pool, uint32_t stack_entries, uint32_t n_spheres, uint32_t n_meshes, uint32_t n_elem_tris);
int megakernel_occupancy_per_cu(uint32_t pool, size_t lds_bytes);
hipError_t launch_primary_cull(const TraceParams& P, hipStream_t stream);
hipError_t launch_sky_resolve(const TraceParams& P, const ResolveParams& R, hipStream_t stream);
hipError_t launch_code_load(hipStream_t stream);
hipError_t launch_resolve(const ResolveParams& R, hipStream_t stream);
hipError_t launch_unpack(const float* gathered, uint32_t width, uint32_t height, uint32_t world,
                         size_t rank_stride_pixels, float* out_radiance, uint8_t* out_rgb8, hipStream_t stream);
hipError_t launch_trace_rays(const TraceParams& P, const float* rays, size_t n, float* out_t, int32_t* out_obj,
                             int32_t* out_tri, float* out_dist, hipStream_t stream);
hipError_t launch_gate_selftest(const float* d_box, const float* d_rays, size_t n, uint8_t* d_fast, uint8_t* d_exact);
hipError_t launch_ieee_selftest(uint64_t seed, size_t n, unsigned long long* d_counts);
hipError_t launch_scatter_debug(const DevMaterial* d_mats, const float* d_in_dir, const float* d_p, const float* d_normal,
                                 const uint32_t* d_rng, size_t n, float* d_out_dir, uint8_t* d_out_ok, uint32_t* d_out_rng);
uint64_t host_splitmix64(uint64_t x);
}  // namespace rbrt

using namespace rbrt;

namespace {

thread_local std::string g_last_error;
thread_local rbrt_hip_call_times_t g_last_render_times{};  // of this thread's last rbrt_hip_render (rbrt_hip_debug.h)

// GPU_MAX_HW_QUEUES as the process had it when this library was LOADED: the HIP runtime reads the variable when it starts,
// and a host that exports it later (an embedding application, a profiler's preload that initialised the GPU first) would
// make the library plan for eight queues on a runtime that has four.
const uint32_t g_hw_queues_at_load = [] {
    if (const char* q = std::getenv("GPU_MAX_HW_QUEUES")) {
        char* end = nullptr;
        const long v = std::strtol(q, &end, 10);
        if (end != q && *end == '\0' && v >= 1 && v <= 64) return uint32_t(v);
    }
    return 4u;  // the runtime's default
}();

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            int _code = (_e == hipErrorOutOfMemory) ? RBRT_ERR_OOM                                   \
                        : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? RBRT_ERR_NO_DEVICE \
                                                                                  : RBRT_ERR_HIP;    \
            return fail(_code, std::string(#expr) + ": " + hipGetErrorString(_e));                   \
        }                                                                                            \
    } while (0)

uint32_t local_tiles_of(uint32_t n_tiles, uint32_t rank, uint32_t world) {
    return rank < n_tiles ? (n_tiles - rank + world - 1u) / world : 0u;
}

// Lab knobs (include/rbrt_hip_debug.h): environment variables that tune the kernels' scheduling are read only when
// RBRT_HIP_LAB=1, and a value outside its range is an error, not something to clamp silently.
const char* lab_env(const char* name) {
    const char* lab = std::getenv("RBRT_HIP_LAB");
    return (lab && lab[0] == '1') ? std::getenv(name) : nullptr;
}
// dst = $name when set (lab mode); false + message when it is not an integer in [lo, hi] (or not in `allowed`, if given)
bool lab_u32(const char* name, long long lo, long long hi, uint32_t& dst, std::string& err, std::initializer_list<long long> allowed = {}) {
    const char* e = lab_env(name);
    if (!e) return true;
    char* end = nullptr;
    const long long v = std::strtoll(e, &end, 10);
    bool ok = end != e && *end == '\0' && v >= lo && v <= hi;
    if (ok && allowed.size() != 0) ok = std::find(allowed.begin(), allowed.end(), v) != allowed.end();
    if (!ok) {
        err = std::string("lab knob ") + name + "=" + e + " is not valid (see include/rbrt_hip_debug.h)";
        return false;
    }
    dst = uint32_t(v);
    return true;
}

size_t workspace_cap_bytes() {
    size_t mb = 1024;
    if (const char* e = std::getenv("RBRT_HIP_WORKSPACE_MB")) {
        long v = std::atol(e);
        if (v > 0) mb = size_t(v);
    }
    return mb << 20;
}

}  // namespace

constexpr uint32_t kStripesAuto = 0xFFFFFFFFu;  // rbrt_hip_scene::work_stripes_overlap: chosen per launch

struct rbrt_hip_scene {
    int device = 0;
    uint32_t n_spheres = 0, n_meshes = 0, n_elem_tris = 0;
    std::vector<void*> allocs;  // everything hipMalloc'ed for the scene itself
    char* slab_cur = nullptr;   // dev_alloc: the scene's arrays below kSlabMaxPiece are cut from slabs (a hipMalloc / hipFree
    size_t slab_left = 0;       // pair costs 0.1-0.15 ms whatever its size: eleven of them were a tenth of a one-shot call)
    DevSphere* d_spheres = nullptr;
    DevTriangle* d_elem_tris = nullptr;  // BasicTriangle elements (triangle.rs:9-34)
    uint32_t* d_elems = nullptr;         // Scene::elements order (null: no triangles, element e = sphere e)
    DevMaterial* d_materials = nullptr;
    DevMesh* d_meshes = nullptr;
    BvhTri* d_tris = nullptr;  // all meshes' triangle records
    DevCounters* d_counters = nullptr;
    // workspace, grown on demand
    // Frame pipeline: consecutive trace launches (the batches of one render, or successive renders) alternate
    // over `pipeline` lanes. Every lane has its own sample buffer, work counters and per-wave scratch, and --
    // when pipeline > 1 -- its own stream, so that the next launch's waves move in while the previous launch
    // is still finishing its last, poorly filled waves (its drain, up to a third of a short launch). The
    // resolve of a launch stays on the caller's stream, after an event; pipeline == 1 runs everything on the
    // caller's stream.
    struct Lane {
        hipStream_t stream = nullptr;  // null: the caller's stream (pipeline == 1)
        int stream_pool_key = -1;      // >= 0: the stream came from the process's pool of one-shot lane streams and goes back to it
        // ONE sample buffer + work counters per lane, and the lane's next launch waits for the resolve of its previous one.
        // (Round 4 gave every lane a second set, taken in turn, so that the next launch could start the moment the previous
        // one ended instead of 90-120 us later -- resolve and background kernel on the caller's stream, a cross-stream hop
        // each way, profiles/r04_trace_eighths.txt. It was SLOWER: pipelined frame 3.60 -> 3.83 ms. The new launch's waves
        // take every slot the old one frees before the resolve is even runnable, and the resolve -- 20 us of work -- then
        // waits hundreds of microseconds for waves of the OTHER launch to exit, with every later resolve queued behind
        // it on the caller's stream. The gap is the window the short kernels run in.)
        struct Buf {
            float* d_sample_buf = nullptr;
            size_t sample_buf_bytes = 0;
            unsigned long long* d_work_counter = nullptr;
            hipEvent_t ev_resolved = nullptr;
            bool in_use = false;  // ev_resolved has been recorded at least once
        } bufs[1];
        uint32_t* d_gseq = nullptr;
        uint32_t* d_gstack = nullptr;
        // The tile pass (kernels.hip primary_cull_kernel + tile_lists_kernel): which of the rank's tiles see only the
        // background, for one camera and tile partition (`key`). A lane has TWO sets of tables: the pass for a camera the
        // lane has not seen is issued when the CALL is made, on the scene's high-priority `prep_stream`, into the set the
        // lane's launch in flight is not reading -- so it runs beside the other lanes' launches, one or two frames ahead of
        // the trace launch that needs it, instead of in front of that launch on the lane's own stream with half the
        // GPU waiting (a new camera per frame cost 3.77 ms per frame against 3.55 for a fixed one).
        struct TileKey {
            rbrt_camera_t cam;
            uint32_t rank, world;
            uint32_t list_mode;  // the order of the work list (tile_lists_kernel): by the kind of launch, list_mode_for()
            float min_dist;      // the tree boxes of the pass are grown by a pad that holds 1 / min_dist (TraceParams::eps_frac)
        };
        struct TileSet {
            uint32_t* d_cull = nullptr;    // [n_tiles]
            uint32_t* d_lists = nullptr;   // [kTileListHeader + 2 * n_local]
            hipEvent_t ev_lists = nullptr;  // recorded behind the tile pass that filled the set
            hipEvent_t ev_free = nullptr;   // recorded behind the last reader (the resolve kernels of the launch that used it)
            bool free_recorded = false;
            bool key_valid = false;
            TileKey key;
            uint64_t last_used = 0;        // launch number (scene-wide) of the last launch that read the set
        } tiles[2];
        size_t tile_cull_words = 0, tile_lists_words = 0;
        hipEvent_t ev_traced = nullptr;
        bool in_use = false;  // ev_traced has been recorded at least once
        // Elastic launches (below): the lane's helper words on the device ([0] helper waves holding work, [1] number of the
        // last resolved launch), the number of its last launch, the stream and event of its helper launches, and what the
        // watcher needs to issue one for the launch in flight.
        uint32_t* d_helper_words = nullptr;
        uint32_t seq = 0;
        hipEvent_t ev_helper = nullptr;   // behind the last helper launch of THIS lane's launch (whatever stream carried it)
        hipEvent_t ev_carried = nullptr;  // behind the last helper launch this lane's stream carried for another lane
        bool helper_carried = false;
        bool helper_pending = false;  // a helper launch has been issued since the lane's last launch: the next one waits for it
        hipEvent_t ev_ready = nullptr;  // recorded on the lane's stream in front of every launch: everything the launch waits for has happened
        struct Open {                   // the lane's launch in flight, as the watcher needs it
            bool valid = false;
            TraceParams P;
            uint32_t grid = 0, helper_waves = 0, rounds = 0;
            uint64_t seq = 0;           // issue number: a later launch has more of its work left
            bool share = false;
        } open;
    };
    std::vector<Lane> lanes;
    // Elastic launches: the watcher thread and what it shares with the caller's thread.
    std::mutex mu;                 // the caller's thread holds it inside every render call, the watcher while it looks and issues
    std::condition_variable cv;
    std::thread watcher;
    bool watcher_stop = false;
    double last_call_s = 0.0;      // when the caller last issued a launch
    uint64_t open_seq = 0;         // Lane::Open::seq of the launch issued last
    uint32_t helpers_mode = 1;     // RBRT_HELPERS (lab): 0 never, 1 when the GPU has room and the caller has stopped issuing, 2 with every launch (tests)
    uint32_t n_helper_launches = 0;  // since set_timing(1)
    // RBRT_HELPER_MIN_ITEMS (lab): a helper wave joins a launch only while this many work items per wave are left. A wave
    // does ~1,700 items per millisecond in the bulk of the headline frame; twenty frames between fences, ten rounds in one
    // process (profiles/r05_helpers_threshold_sweep.txt), median / worst ms per step: no helpers 3.582 / 3.631; 1536: 3.506 /
    // 3.813 (three rounds of ten WORSE than without: late joiners turn bulk into drain); 4096: 3.513 / 3.544; 8192: 3.536 / 3.578
    uint32_t helper_min_items = 4096;
    uint32_t helper_min_free_per_cu = 1;  // RBRT_HELPER_MIN_FREE (lab): helper launches only while this many wave slots per CU are free
    uint32_t helper_min_launch_mi = 16;  // RBRT_HELPER_MIN_LAUNCH_MI (lab): launches of this many Mi work items or more are helped
    uint32_t helper_rounds = 4;        // RBRT_HELPER_ROUNDS (lab)
    hipStream_t prep_stream = nullptr;  // high priority: the tile passes of cameras the lanes have not seen
    bool one_shot = false;              // made by rbrt_hip_render for one render (CreateHint::one_shot)
    hipStream_t aux_stream = nullptr;   // RBRT_HELPERS=2 (tests): carries the helper launches
    void* h_zeros = nullptr;            // pinned: what a lane's work counters are set to again behind a helper launch
    uint64_t launch_no = 0;
    bool streaming_hint = false;  // the last trace launch was issued while another one was still running
    uint32_t pipeline = 0;   // RBRT_PIPELINE / rbrt_hip_scene_set_pipeline; 0 = automatic (depth_for)
    uint32_t scratch_waves = 0;
    uint32_t n_cus = 256;
    uint32_t hw_queues = 4;  // hardware queues the HIP runtime maps streams onto: $GPU_MAX_HW_QUEUES as scene_create finds it, else the runtime's 4
    bool waves_fixed = false;  // RBRT_WAVES_PER_CU given: no automatic half-size grids
    uint32_t overlap_waves_per_cu = 0;  // RBRT_OVERLAP_WAVES_PER_CU: waves per CU of a launch issued while another is running (0: automatic)
    uint32_t next_lane = 0;
    float* d_acc = nullptr;
    size_t acc_bytes = 0;
    bool poison_samples = false;      // RBRT_POISON_SAMPLES
    // host copy of the objects' bounds, for the tile-direction choice below
    struct Bound {
        float lo[3], hi[3];
    };
    std::vector<Bound> h_bounds;
    uint32_t stack_need = 1;  // deepest BVH: 3 per level + 1
    uint32_t n_waves = 0;  // persistent megakernel grid: as many single-wave workgroups as fit the LDS
    uint32_t pool = 128;          // path slots per wave (RBRT_POOL = 128 | 256)
    uint32_t stack_entries = kLdsStack;  // per-lane stack entries kept in LDS (RBRT_LDS_STACK)
    uint32_t y_low_water = 28;    // RBRT_Y_LOW
    uint32_t y_high_water = 28, y_high_min_parked = 16;  // RBRT_Y_HIGH, RBRT_Y_HIGH_PARKED
    uint32_t leaf_round = 6;      // RBRT_LEAF_ROUND
    uint32_t leaf_leaves = 16;    // RBRT_LEAF_LEAVES
    uint32_t share_idle = 4;      // RBRT_SHARE_IDLE (0: no shared traversals)
    uint64_t share_below = ~0ull;  // RBRT_SHARE_BELOW: launches under this many samples use the sharing build (all)
    uint32_t drain_mode = 1;      // RBRT_DRAIN_MODE
    uint32_t work_stripes = 16;   // RBRT_WORK_STRIPES: chunks (of 64 work items) per stripe for a launch that has the GPU to itself; 0 = contiguous shards
    // RBRT_WORK_STRIPES_OVERLAP: the same for a launch issued while another is running. Automatic (kStripesAuto): contiguous
    // shards for a big launch, stripes of 4 chunks for one below 20 M work items -- measured per step on one box, 4 against
    // 0: the frame of config 2 (39 M) +1.0 %, the 871k mesh +2.0 %, the rough mesh -0.8 %; a half (20 M) -0.8 %, a quarter
    // -2.7 %, an eighth -2.8 %, a 512x384 frame -2.5 %
    uint32_t work_stripes_overlap = kStripesAuto;
    uint32_t tile_classes = 0;    // RBRT_TILE_CLASSES (order of the work list by tile class, tile_lists_kernel)
    bool tile_classes_set = false;  // ... given: it overrides the rule of list_mode_for()
    bool trace_launches = false;    // RBRT_TRACE_LAUNCHES=1 (lab): one stderr line per trace launch and per tile pass
    uint32_t isolated_list_mode = 4;  // RBRT_TILE_ISOLATED_MODE: the list mode of a launch that has the GPU to itself
    uint32_t tile_tail_div = 8;       // RBRT_TILE_TAIL_DIV (mode 4)
    uint32_t tile_order = 0;      // RBRT_TILE_ORDER (0: empty_end_is_first decides; 1: first-to-last; 2: last-to-first)
    uint32_t primary_cull = 1;    // RBRT_PRIMARY_CULL (0: no tile pass, the trace kernel renders every tile)
    uint32_t shade_rounds = 1;    // RBRT_SHADE_ROUNDS (rounds while work items are left; unbounded afterwards)
    uint32_t shade_cont_min = 8;  // RBRT_SHADE_CONT_MIN
    // stats / timing
    rbrt_hip_stats_t stats{};
    bool stats_pending = false;
    bool timing = false;
    std::vector<hipEvent_t> events;  // [t0, t1, r1] per batch
    size_t events_used = 0;
    bool timing_overflow = false;  // more than kMaxTimedLaunches launches since set_timing: the rest go untimed
    uint32_t last_batch = 0, last_n_batches = 0;  // sample batching of the last render (rbrt_hip_scene_last_batching)
    uint32_t n_full_grid = 0, n_half_grid = 0;    // trace launches since set_timing(1), by grid size (rbrt_hip_scene_kernel_ms_ex)
    uint64_t total_nodes = 0, total_tris = 0;
    uint32_t n_device_built = 0;  // meshes whose BVH the GPU built
    rbrt_hip_call_times_t create_times{};  // where scene_create's time went (rbrt_hip_scene_create_times)
    std::unique_ptr<struct Refine> refine;  // the host builder's tree of device-built meshes, made in the background (below)
    uint32_t n_refined = 0;                 // meshes whose device-built tree has been replaced by it
};

// ---- The tree a scene STARTS with and the tree it goes on with -------------------------------------------------------
// rbrt_hip_scene_create cannot know how long its handle will live: the reference's one call (src/main.rs:82) renders a
// 1024x768x50 frame in 4 ms here, and the host's binned-SAH build of the bunny-sized mesh costs twenty times that, while
// the device builder makes a tree in about a millisecond per hundred thousand triangles whose frames are 2-3 % slower
// (clustered over the Morton order). So a mesh the device builder accepts gets its tree FIRST,
// the handle is usable at once, and a background thread makes the SAH tree from the very records the device builder
// emitted (read back from the device: the caller's arrays are only borrowed for the duration of scene_create) and uploads
// it into fresh allocations; the first render call that finds it ready switches the scene's pointers over. Launches
// in flight keep reading the old arrays, which live until the handle is destroyed. Any tree with bvh.h's properties gives
// the scan's answer, so the image does not change (tests/test_bvh_refine.py). rbrt_hip_scene_destroy cancels a build
// nobody waits for; the one-shot rbrt_hip_render does not start one. $RBRT_BVH_REFINE=0 turns it off, RBRT_BVH_BUILDER =
// host | device force one builder (and no refinement).
struct Refine {
    std::thread th;
    std::atomic<int> state{0};  // 0: the thread is at work; 1: results ready, not adopted yet; 2: nothing (more) to adopt
    std::atomic<bool> cancel{false};
    // in
    int device = 0;
    const BvhTri* d_tris = nullptr;  // the scene's record array as the device builder left it
    size_t tri_total = 0;
    std::vector<DevMesh> meshes;     // the scene's mesh table
    std::vector<uint32_t> tri_base, n_valid;
    std::vector<uint8_t> wanted;     // per mesh: 1 = device-built, to be rebuilt
    unsigned max_threads = 0;
    // out
    std::vector<void*> allocs;
    BvhTri* d_tris_new = nullptr;
    DevMesh* d_meshes_new = nullptr;
    uint32_t stack_need = 1, n_done = 0;
    uint64_t nodes_delta_plus = 0, nodes_delta_minus = 0;
    double seconds = 0.0;  // start of the thread to results on the device
    std::string error;

    void run() {
        const double t0 = now_s();
        hipStream_t st = nullptr;
        std::vector<BvhTri> h;
        const auto finish = [&](const std::string& why) {
            if (st) (void)hipStreamDestroy(st);
            if (!why.empty()) {
                for (void* p : allocs) (void)hipFree(p);
                allocs.clear();
                error = why;
            }
            seconds = now_s() - t0;
            state.store(why.empty() ? 1 : 2, std::memory_order_release);
        };
        const auto hip = [&](hipError_t e, const char* what) -> bool {
            if (e == hipSuccess) return true;
            finish(std::string(what) + ": " + hipGetErrorString(e));
            return false;
        };
        if (!hip(hipSetDevice(device), "hipSetDevice")) return;
        if (!hip(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate")) return;
        h.resize(tri_total);
        if (!hip(hipMemcpyAsync(h.data(), d_tris, tri_total * sizeof(BvhTri), hipMemcpyDeviceToHost, st), "read-back of the records")) return;
        if (!hip(hipStreamSynchronize(st), "read-back of the records")) return;
        std::vector<BvhBuildResult> built(meshes.size());
        BvhBuildOptions opt;
        opt.max_threads = max_threads, opt.cancel = &cancel;
        for (size_t i = 0; i < meshes.size(); ++i) {
            if (!wanted[i]) continue;
            built[i] = build_bvh_from_records(h.data() + tri_base[i], n_valid[i], opt);
            if (built[i].cancelled || cancel.load()) return finish("cancelled");
            BvhBuildResult& b = built[i];
            const size_t cap = size_t(i + 1 < meshes.size() ? tri_base[i + 1] : uint32_t(tri_total)) - tri_base[i];
            if (b.tris.size() > cap || b.nodes.empty()) return finish("the host builder's tree does not fit the mesh's records");
            if (tri_base[i] != 0)
                for (BvhNode4& nd : b.nodes)
                    for (int c = 0; c < 4; ++c)
                        if (nd.child[c] < 0 && nd.child[c] != kNoChild) {
                            const uint32_t leaf = uint32_t(~nd.child[c]);
                            nd.child[c] = ~int32_t((((leaf >> kLeafBits) + tri_base[i]) << kLeafBits) | (leaf & uint32_t(kLeafMax - 1)));
                        }
            void* p = nullptr;
            if (!hip(hipMalloc(&p, b.nodes.size() * sizeof(BvhNode4)), "hipMalloc(nodes)")) return;
            allocs.push_back(p);
            if (!hip(hipMemcpyAsync(p, b.nodes.data(), b.nodes.size() * sizeof(BvhNode4), hipMemcpyHostToDevice, st), "upload of the nodes")) return;
            std::copy(b.tris.begin(), b.tris.end(), h.begin() + tri_base[i]);
            nodes_delta_minus += meshes[i].n_nodes, nodes_delta_plus += b.nodes.size();
            meshes[i].nodes = static_cast<const BvhNode4*>(p);
            meshes[i].max_e12 = b.max_e12;
            meshes[i].n_nodes = uint32_t(b.nodes.size());
            meshes[i].n_tris = uint32_t(b.tris.size());
            stack_need = std::max(stack_need, b.stack_need);
            ++n_done;
        }
        void* p = nullptr;
        if (!hip(hipMalloc(&p, std::max<size_t>(tri_total * sizeof(BvhTri), 64)), "hipMalloc(records)")) return;
        allocs.push_back(p);
        d_tris_new = static_cast<BvhTri*>(p);
        if (!hip(hipMemcpyAsync(p, h.data(), tri_total * sizeof(BvhTri), hipMemcpyHostToDevice, st), "upload of the records")) return;
        for (DevMesh& dm : meshes) dm.tris = d_tris_new;
        if (!hip(hipMalloc(&p, std::max<size_t>(meshes.size() * sizeof(DevMesh), 16)), "hipMalloc(mesh table)")) return;
        allocs.push_back(p);
        d_meshes_new = static_cast<DevMesh*>(p);
        if (!hip(hipMemcpyAsync(p, meshes.data(), meshes.size() * sizeof(DevMesh), hipMemcpyHostToDevice, st), "upload of the mesh table")) return;
        if (!hip(hipStreamSynchronize(st), "upload of the refined trees")) return;
        finish("");
    }
};

namespace {

int ensure_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RBRT_ERR_NO_DEVICE, std::string("no HIP device available: ") +
                                            (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= n) return fail(RBRT_ERR_INVALID_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    return RBRT_OK;
}

// Device memory that lives as long as the scene (released by rbrt_hip_scene_destroy, never one by one). Called from the
// thread that holds the scene (scene_create, or a render call under s->mu); the background builder has allocations of its own.
constexpr size_t kSlabBytes = 32u << 20, kSlabMaxPiece = 8u << 20, kSlabAlign = 512;
hipError_t dev_alloc(rbrt_hip_scene* s, size_t bytes, void** out) {
    bytes = (std::max<size_t>(bytes, 1) + kSlabAlign - 1) & ~(kSlabAlign - 1);
    *out = nullptr;
    if (bytes > kSlabMaxPiece) {
        const hipError_t e = hipMalloc(out, bytes);
        if (e == hipSuccess) s->allocs.push_back(*out);
        return e;
    }
    if (bytes > s->slab_left) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, kSlabBytes);
        if (e != hipSuccess) return e;
        s->allocs.push_back(p);
        s->slab_cur = static_cast<char*>(p), s->slab_left = kSlabBytes;
    }
    *out = s->slab_cur;
    s->slab_cur += bytes, s->slab_left -= bytes;
    return hipSuccess;
}

template <class T>
int upload(rbrt_hip_scene* s, const std::vector<T>& host, T** out) {
    void* d = nullptr;
    size_t bytes = std::max<size_t>(host.size() * sizeof(T), 16);
    HIP_TRY(dev_alloc(s, bytes, &d));
    if (!host.empty()) HIP_TRY(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<T*>(d);
    return RBRT_OK;
}

// Uploads a mesh's SoA arrays and builds its BVH on the GPU (bvh_device.hip): triangle records to d_tris_out (leaf
// links absolute, d_tris_out[0] being record tri_base of the scene's array), normals to d_normals (may be null).
// r->ok says whether a tree was built; on error nothing is left allocated.
hipError_t device_build_mesh(const rbrt_mesh_t& m, BvhTri* d_tris_out, uint32_t tri_base, Normal4* d_normals, DeviceBvhResult* r,
                             double* upload_s = nullptr) {
    const double t_up0 = now_s();
    const float* src[12] = {m.v0x, m.v0y, m.v0z, m.e1x, m.e1y, m.e1z, m.e2x, m.e2y, m.e2z, m.nx, m.ny, m.nz};
    float* d_soa = nullptr;
    uint8_t* d_pad = nullptr;
    const size_t stride = (size_t(m.n_total) + 63u) & ~size_t(63);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_soa), stride * 12u * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_pad), m.n_total);
    for (int k = 0; k < 12 && e == hipSuccess; ++k)
        e = hipMemcpy(d_soa + stride * k, src[k], size_t(m.n_total) * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_pad, m.is_padding, m.n_total, hipMemcpyHostToDevice);
    if (upload_s) *upload_s = now_s() - t_up0;
    if (e == hipSuccess) {
        const DeviceMeshSoa soa = {d_soa, d_soa + stride, d_soa + 2 * stride, d_soa + 3 * stride, d_soa + 4 * stride,
                                   d_soa + 5 * stride, d_soa + 6 * stride, d_soa + 7 * stride, d_soa + 8 * stride, d_pad};
        int algo = 0;
        if (const char* a = lab_env("RBRT_BVH_DEVICE_ALGO")) algo = !std::strcmp(a, "lbvh") ? 1 : 0;
        e = build_bvh_device(soa, m.n_total, d_tris_out, tri_base, r, nullptr, algo);
    }
    if (e == hipSuccess && r->ok && d_normals)
        e = device_normals(d_soa + 9 * stride, d_soa + 10 * stride, d_soa + 11 * stride, m.n_total, d_normals, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_soa), (void)hipFree(d_pad);
    if (e != hipSuccess && r->d_nodes) {
        (void)hipFree(r->d_nodes);
        r->d_nodes = nullptr, r->ok = false;
    }
    return e;
}

uint32_t div_magic_of(uint32_t d) { return d <= 1 ? 0xFFFFFFFFu : uint32_t(0x100000000ull / d); }  // kernels.hip div_magic

constexpr uint32_t kMaxPipeline = 8;
constexpr uint32_t kLanesAtCreate = 4;  // lanes made by scene_create (deeper pipelines: rbrt_hip_scene_set_pipeline makes the rest)
constexpr size_t kMaxTimedLaunches = 4096;  // timing events are recycled per set_timing, never more than this many launches

// Depth of the frame pipeline: what was set, or 8 (4 on the runtime's default four hardware queues). Four or five part-grid
// launches are resident at any time (grid_for); the lanes beyond them hold launches that are READY the moment a resident one drains. For a short launch the time between its end and the
// moment its lane's next launch can start (resolve, background kernel, two cross-stream hops: ~0.1 ms) is a fifth of a
// step, and the extra lanes hide it. Measured with a new camera every frame (round 4, tools/ab_knobs.sh, per step):
//   depth              3       4       5       6       8
//   frame             3.50    3.55    3.76    3.58    3.50
//   an eighth of it   0.64    0.57    0.74    0.66    0.545      (HIP runtime's default: 4 hardware queues)
//   frame             3.50            3.51    3.50    3.48
//   an eighth         0.64    0.567   0.552   0.545   0.527      (GPU_MAX_HW_QUEUES=8)
// The lanes' streams map onto the runtime's hardware queues: with four of them a fifth and a sixth lane share a queue with
// another lane's launches and barriers, unevenly (5 and 6 deep are WORSE than 4), eight lanes pair up evenly and behave like
// four with more buffers; with eight queues deeper is monotonically better. So: 8 deep, and GPU_MAX_HW_QUEUES=8 is what
// an integrator should export before the HIP runtime starts (bench.py and the rbrt CLI do). Before round
// 4's one-wave short kernels a fourth lane only added a lane waiting for its starved resolve (round 3: "deeper is worse").
// (Every lane has a sample buffer of its own: the lanes' buffers together stay within 8 GiB -- eight lanes at the default
// workspace cap of 1 GiB --, but there are never fewer than three: a caller who raises RBRT_HIP_WORKSPACE_MB gets long
// launches, which need no deep pipeline.)
// A process that has NOT exported GPU_MAX_HW_QUEUES=8 runs on four hardware queues: four launches at most run side by
// side, and the pipeline is four deep with grids to match (grid_for) -- eight lanes of 3 waves per CU each would leave a
// quarter of the wave slots empty (measured with four queues, per frame / per eighth: 8 deep x 3 per CU 4.58 / 0.72 ms;
// 4 deep x 6 per CU 3.57 / 0.578; 8 deep x 6 3.61 / 0.58; 8 deep x 8 3.56 / 0.596; with eight queues 8 deep x 3: 3.42 / 0.47).
uint32_t depth_for(const rbrt_hip_scene* s, size_t sample_buffer_bytes) {
    if (s->pipeline != 0) return s->pipeline;
    const size_t fit = (size_t(8) << 30) / (sample_buffer_bytes ? sample_buffer_bytes : 1);
    if (s->hw_queues < 8u) return fit >= 4 ? 4u : 3u;
    return fit >= 8 ? 8u : fit >= 4 ? 4u : 3u;  // (never 5 to 7: with the runtime's four hardware queues they are worse than 4)
}

// Is a trace launch of this scene still running on another lane? (What decides how the next one is issued.)
bool other_launch_in_flight(const rbrt_hip_scene* s, const rbrt_hip_scene::Lane* mine) {
    for (const auto& L : s->lanes)
        if (&L != mine && L.in_use && L.ev_traced && hipEventQuery(L.ev_traced) == hipErrorNotReady) {
            // (hipErrorNotReady is an answer, not a failure: keep it out of the next launch check -- and clear the error
            // state only then, so that an unrelated sticky error still reaches whoever checks next)
            (void)hipGetLastError();
            return true;
        }
    return false;
}

// Waves of one trace launch. A launch that finds the GPU idle (a blocking caller, the first frame of a stream) takes all
// resident wave slots. One issued while another is still running -- consecutive frames or sample batches queued back to
// back -- takes a PART of them, sized so that the lanes of the pipeline together ask for one and a half times the slots there
// are: 3 of a CU's 16 with eight lanes, 6 with four, 8 with three. The launches then run side by side, each wave works
// through more items before its pool runs empty (the drain's share of a launch shrinks with the grid), and what one
// launch's drain frees the waves of the queued ones take at once. Measured, eight lanes, new camera every frame, waves per
// CU of an overlapped launch -> ms per frame / half / quarter / eighth of config 2 (tools/ab_knobs.sh):
//    8 (rounds 2-3: "half the slots")   3.47   1.80   0.945   0.509
//    4                                  3.44   1.78   0.905   0.478
//    3                                  3.43   1.776  0.900   0.475     (rough stand-in 5.25 -> 5.20, visible 871k mesh 4.28 -> 4.23)
//    2 (all lanes resident, none queued) 3.46  1.825  0.967   0.547
//    1                                  5.93   3.11   1.61    0.87
// Twelve lanes of 2 or 3 waves per CU on sixteen hardware queues: the same as eight of 3 (3.42-3.43 / 0.470-0.471); sixteen: worse.
// (Rounds 2-3 had found smaller grids and deeper pipelines worse than half grids three deep: every launch's resolve then
// waited for a draining launch, DESIGN.md section 6 "Short kernels beside a persistent one".)
// A STREAM has a beginning and an end: its first launches start together and drain together, its last ones are left with
// their small share of an emptying GPU. With 20 frames between two fences instead of 100 (bench.py --steps; tools/k20_sweep.sh),
// ms per frame / half / quarter / eighth:   3 per CU  3.65  1.906  0.996  0.548   (100 frames: 3.431  1.745  0.894  0.470)
//                                            4         3.58  1.857  0.971  0.552               3.437  1.759  0.906  0.486
//                                            6         3.58  1.846  0.990  0.558               3.443  1.776  0.929  0.507
//                                            8         3.54  1.859  0.992  0.588               3.474  1.813  0.949  0.524
// so a launch of 8 M work items or more takes 4 of a CU's slots, a smaller one 3. (Sizing a stream's launches by the number
// in flight when each is issued -- its first ones large, 12, 8, 6 ... -- makes the beginning worse: 3.74 for 20 frames; the
// full grid for a stream's first launch into an idle GPU: 3.49 against 3.59 in one session, 3.66 against 3.55 in the next --
// twenty frames between two fences come out 2-3 % apart from one box and session to the next.)
// `company` (a call that found the GPU idle: `stream` false): the sample batches of the call still to be issued behind this
// one -- the last batch ends alone, the one before it shares with one, ...: with the stream's 3 slots per CU the second batch
// of a blocking 1920x1080x512 eighth ran on 3/16 of the GPU once the first had ended, 38.6 ms for the frame against 14.7.
uint32_t grid_for(const rbrt_hip_scene* s, bool overlapped, uint32_t depth, bool stream, uint32_t company, uint64_t n_items) {
    if (s->waves_fixed || !overlapped) return s->n_waves;
    uint32_t per_cu = s->overlap_waves_per_cu;  // (lab knob; 0: by the launches that run side by side)
    uint32_t side_by_side = depth < s->hw_queues ? depth : s->hw_queues;  // launches that can run at the same time
    if (!stream && company + 1u < side_by_side) side_by_side = company + 1u;
    if (per_cu == 0u) {
        per_cu = side_by_side >= 2u ? (24u + side_by_side - 1u) / side_by_side : 16u;
        if (stream && per_cu < 4u && n_items >= (8ull << 20)) per_cu = 4u;
    }
    const uint32_t part = s->n_cus * per_cu;
    return part < s->n_waves ? part : s->n_waves;
}

// In which order a launch's tiles are handed out (tile_lists_kernel's modes), by what KIND of launch it is -- a rule, not a
// setting made on one frame (measured on five workloads, DESIGN.md section 6 "In which order the work list is handed out"):
// a launch that has the GPU to itself (a blocking caller) ends with every wave finishing what it holds and nothing behind
// it, so its HEAVY tiles -- a mesh box or more than one sphere in reach -- go out first and the launch drains on light
// ones (mode 1: isolated launch of the 871k mesh 6.36 -> 5.00 ms, the rough stand-in 6.31 -> 6.11, an eighth 1.21 -> 1.11;
// the example frame, whose row-major order happens to end on its thin horizon band, 4.04 -> 4.17); a launch issued into
// a stream of launches keeps the image's row-major order (mode 0: within 1.5 % of the best order on all five, and the
// best on three -- the next launch's bulk fills whatever the drain leaves).
uint32_t list_mode_for(const rbrt_hip_scene* s, bool overlapped) {
    if (s->tile_classes_set) return s->tile_classes;
    return overlapped ? 0u : s->isolated_list_mode;
}

// The lane streams of ONE-SHOT scenes (rbrt_hip_render: a scene made for one render and destroyed) are kept by the process and
// handed from call to call: creating and destroying a stream costs each call 0.3-0.4 ms of its 9. Keyed by device and priority;
// a stream goes back idle (rbrt_hip_scene_destroy synchronises the device first) and is never destroyed.
std::mutex g_stream_pool_mu;
std::vector<std::pair<int, hipStream_t>> g_stream_pool;
hipStream_t take_pooled_stream(int key) {
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    for (size_t i = 0; i < g_stream_pool.size(); ++i)
        if (g_stream_pool[i].first == key) {
            const hipStream_t st = g_stream_pool[i].second;
            g_stream_pool.erase(g_stream_pool.begin() + long(i));
            return st;
        }
    return nullptr;
}
void give_pooled_stream(int key, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    g_stream_pool.emplace_back(key, st);
}

// Brings s->lanes to `depth` entries (streams, events, counters and per-wave scratch of each lane).
int ensure_lanes(rbrt_hip_scene* s, uint32_t depth) {
    const size_t had = s->lanes.size();
    while (s->lanes.size() < depth) {
        rbrt_hip_scene::Lane L;
        void* p = nullptr;
        constexpr size_t counter_bytes = sizeof(unsigned long long) * kWorkShards * kWorkCounterStride;
        // (zeroed by a copy, not a memset: a process's first hipMemset loads the runtime's fill kernels, 20 ms in the CLI's
        // one render; the per-wave scratch -- 50 MB per lane -- is made when the lane first gets a launch, size_lane)
        static const unsigned long long zeros[kWorkShards * kWorkCounterStride] = {};
        for (auto& B : L.bufs) {
            HIP_TRY(dev_alloc(s, counter_bytes, &p));
            B.d_work_counter = static_cast<unsigned long long*>(p);
            HIP_TRY(hipMemcpy(p, zeros, counter_bytes, hipMemcpyHostToDevice));
        }
        HIP_TRY(dev_alloc(s, 64, &p));
        L.d_helper_words = static_cast<uint32_t*>(p);
        HIP_TRY(hipMemcpy(p, zeros, 64, hipMemcpyHostToDevice));
        // the lane is recorded before its stream and events exist, so that a failure below leaves them to
        // rbrt_hip_scene_destroy instead of leaking them (a lane without a stream is never selected: the caller
        // gets the error)
        s->lanes.push_back(L);
        rbrt_hip_scene::Lane& R = s->lanes.back();
        // The lanes' streams carry the long trace launches and nothing else. They get the LOWEST priority: the short
        // kernels that wait on them -- the resolve on the caller's stream, a collective's copy kernels, the unpack --
        // then win the wave slots a finishing trace wave frees, ahead of the next trace launch's waves.
        int prio_low = 0, prio_high = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);  // (numerically: low >= high)
        if (const char* pe = lab_env("RBRT_LANE_PRIORITY")) prio_low = !std::strcmp(pe, "default") ? 0 : !std::strcmp(pe, "high") ? prio_high : prio_low;
        hipError_t e = hipSuccess;
        if (s->one_shot) {
            R.stream_pool_key = s->device * 64 + (prio_low & 63);
            R.stream = take_pooled_stream(R.stream_pool_key);
        }
        if (!R.stream) e = hipStreamCreateWithPriority(&R.stream, hipStreamNonBlocking, prio_low);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&R.ev_helper, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&R.ev_carried, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&R.ev_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&R.ev_traced, hipEventDisableTiming);
        for (auto& B : R.bufs)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&B.ev_resolved, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (R.stream && R.stream_pool_key >= 0) give_pooled_stream(R.stream_pool_key, R.stream);
            else if (R.stream) (void)hipStreamDestroy(R.stream);
            if (R.ev_helper) (void)hipEventDestroy(R.ev_helper);
            if (R.ev_carried) (void)hipEventDestroy(R.ev_carried);
            if (R.ev_ready) (void)hipEventDestroy(R.ev_ready);
            if (R.ev_traced) (void)hipEventDestroy(R.ev_traced);
            for (auto& B : R.bufs)
                if (B.ev_resolved) (void)hipEventDestroy(B.ev_resolved);
            s->lanes.pop_back();
            return fail(RBRT_ERR_HIP, std::string("pipeline lane: ") + hipGetErrorString(e));
        }
    }
    (void)had;  // (the counters were zeroed by blocking copies: nothing of the new lanes is in flight)
    return RBRT_OK;
}

// ---- Elastic launches ------------------------------------------------------------------------------------------------
// A launch's grid is fixed when it is issued -- a part of the wave slots while launches overlap (grid_for) -- but whether
// the GPU stays full is decided later: when the caller stops issuing (the end of a stream of frames; an application that
// keeps only two or three frames in flight) the launches still running keep their part of an emptying GPU: 3.5 ms of the
// 70 ms that twenty frames between two fences take (profiles/r04_trace_frames.txt). The kernel hands its work out from
// counters in memory, so a launch can be given more waves at any time: a HELPER launch of the same kernel with the same
// parameters draws from the same counters into the same sample buffer. What has to be arranged is that nobody reads the
// samples before a helper wave that holds some has written them, and that a helper wave arriving after the launch is over
// takes nothing from counters that by then belong to the lane's next launch:
//   * a helper wave adds itself to helper_words[0] BEFORE its first draw and leaves after its last store (release); the
//     launch's resolve, which runs after the launch itself has ended -- nobody can draw any more --, waits for the count to
//     return to zero (bounded; megakernel.inl, kernels.hip resolve_kernel);
//   * the resolve then publishes the launch's number in helper_words[1] and only then resets the counters; a helper
//     wave draws only while the published number is still the PREVIOUS launch's. One that looked just before the number
//     changed may take one more chunk from the reset counters: it renders samples of its own launch once more -- the same
//     bits -- into a buffer nobody else writes, because the lane's next launch waits for the helper LAUNCH to end
//     (ev_helper) and zeroes the counters again before it starts.
// Who issues them: a watcher thread (one per scene handle that has seen a stream of calls), which every 100 us retires
// the launches that have ended and, once the caller has not issued anything for 200 us and the launches in flight leave a
// wave per CU or more free, gives the free slots to them -- up to four rounds per launch as others end --, each helper
// launch carried by the stream of a lane that has nothing in flight (issue_helper). The launch itself
// is untouched (the helper is a build of its own), so a stream in full flow pays nothing. RBRT_HELPERS (lab): 0 off, 2 =
// a helper with EVERY overlapped launch (tests: the protocol under every scene of the suite).
// `carrier`: the stream the helper launch is issued on. The runtime maps streams onto a few hardware queues and a queue
// takes its packets in order, so a helper launch on a stream of its own would sit in some lane's queue, behind that lane's
// launch or in front of its next one; the streams of lanes that have nothing in flight are the queues that are free, and
// at the end of a stream of frames there are more of them with every launch that ends.
int issue_helper(rbrt_hip_scene* s, rbrt_hip_scene::Lane& L, uint32_t waves, hipStream_t carrier) {
    if (!L.open.valid || waves == 0u || L.open.grid + L.open.helper_waves + waves > s->scratch_waves) return RBRT_OK;
    HIP_TRY(hipStreamWaitEvent(carrier, L.ev_ready, 0));  // (what the launch waited for: the resolve before it, its tables)
    TraceParams P = L.open.P;
    P.wave_base = L.open.grid + L.open.helper_waves;
    P.helper_min_items = s->helpers_mode == 2u ? 0u : s->helper_min_items;  // (tests: every helper wave joins)
    HIP_TRY(launch_trace_helper(P, waves, s->pool, L.open.share, carrier));
    HIP_TRY(hipEventRecord(L.ev_helper, carrier));
    L.helper_pending = true;
    L.open.helper_waves += waves, L.open.rounds += 1u;
    s->n_helper_launches += 1u;
    if (s->trace_launches)
        std::fprintf(stderr, "[rbrt_hip] t %.3f ms helper launch: lane %u, %u waves behind %u\n", now_s() * 1e3, unsigned(&L - s->lanes.data()), waves, P.wave_base);
    return RBRT_OK;
}

void watcher_main(rbrt_hip_scene* s) {
    (void)hipSetDevice(s->device);
    std::unique_lock<std::mutex> lk(s->mu);
    while (!s->watcher_stop) {
        bool any = false;
        for (const auto& L : s->lanes) any = any || L.open.valid;
        if (!any) {
            s->cv.wait(lk);
            continue;
        }
        lk.unlock();
        std::this_thread::sleep_for(std::chrono::microseconds(100));
        lk.lock();
        if (s->watcher_stop) break;
        uint32_t n_open = 0, resident = 0;
        std::vector<rbrt_hip_scene::Lane*> carriers;
        for (auto& L : s->lanes) {
            if (L.open.valid) {
                const hipError_t q = hipEventQuery(L.ev_traced);
                if (q == hipErrorNotReady) {
                    (void)hipGetLastError();
                    ++n_open, resident += L.open.grid + L.open.helper_waves;
                    continue;
                }
                L.open.valid = false;  // (ended, or an error the caller's next call will meet)
                if (s->trace_launches) std::fprintf(stderr, "[rbrt_hip] t %.3f ms lane %u: its launch has ended\n", now_s() * 1e3, unsigned(&L - s->lanes.data()));
            }
            // a lane with nothing in flight: its stream can carry a helper launch (unless it still carries one)
            if (L.stream && (!L.helper_carried || hipEventQuery(L.ev_carried) == hipSuccess)) carriers.push_back(&L);
            (void)hipGetLastError();
        }
        if (n_open == 0u || carriers.empty() || now_s() - s->last_call_s < 200e-6) continue;
        const uint32_t free_waves = s->n_waves > resident ? s->n_waves - resident : 0u;
        if (free_waves < s->n_cus * s->helper_min_free_per_cu) continue;
        // The free slots go to the launch issued LAST first (it has the most work left: of the launches of a stream's end the
        // last one otherwise ends alone, a frame's time after the others), up to what its scratch holds, then to the one before.
        std::vector<rbrt_hip_scene::Lane*> open_lanes;
        for (auto& L : s->lanes)
            if (L.open.valid) open_lanes.push_back(&L);
        std::sort(open_lanes.begin(), open_lanes.end(), [](const rbrt_hip_scene::Lane* a, const rbrt_hip_scene::Lane* b) { return a->open.seq > b->open.seq; });
        uint32_t left = free_waves;
        for (rbrt_hip_scene::Lane* LP : open_lanes) {
            rbrt_hip_scene::Lane& L = *LP;
            // (a short launch is over before a helper launch has arrived -- the watcher looks every 100 us --: with helpers an
            // eighth of the headline frame, 0.5 ms, came out 2 % SLOWER, a quarter 1 % slower, a half equal, the frame 2 %
            // faster: launches of 16 M work items or more are the ones that are helped)
            if (L.open.rounds >= s->helper_rounds || left < 64u || carriers.empty() || L.open.P.n_items < (uint64_t(s->helper_min_launch_mi) << 20)) continue;
            const uint32_t has = L.open.grid + L.open.helper_waves;
            const uint32_t room = s->scratch_waves > has ? (s->scratch_waves - has) / 64u * 64u : 0u;
            const uint32_t w = std::min(left / 64u * 64u, room);
            if (w < 64u) continue;
            rbrt_hip_scene::Lane* C = carriers.back();
            carriers.pop_back();
            if (issue_helper(s, L, w, C->stream) != RBRT_OK) break;  // (the caller's next call reports what is wrong with the device)
            if (hipEventRecord(C->ev_carried, C->stream) != hipSuccess) break;
            C->helper_carried = true;
            left -= w;
        }
    }
}

// Switches the scene over to the background thread's trees once they are on the device (struct Refine). Launches issued
// before keep the arrays they were given; nothing is freed before rbrt_hip_scene_destroy.
void adopt_refined(rbrt_hip_scene* s) {
    Refine* r = s->refine.get();
    if (!r || r->state.load(std::memory_order_acquire) != 1) return;
    if (r->th.joinable()) r->th.join();
    s->d_tris = r->d_tris_new;
    s->d_meshes = r->d_meshes_new;
    s->allocs.insert(s->allocs.end(), r->allocs.begin(), r->allocs.end());
    r->allocs.clear();
    s->stack_need = std::max(s->stack_need, r->stack_need);
    s->total_nodes = s->total_nodes + r->nodes_delta_plus - r->nodes_delta_minus;
    s->n_device_built -= std::min(s->n_device_built, r->n_done);
    s->n_refined += r->n_done;
    r->state.store(2, std::memory_order_release);
}

int check_material(const rbrt_material_t& m) {
    if (m.kind < RBRT_MAT_LAMBERTIAN || m.kind > RBRT_MAT_DIELECTRIC)
        return fail(RBRT_ERR_INVALID_ARG, "unknown material kind");
    return RBRT_OK;
}

int fill_trace_params(const rbrt_hip_scene* s, const rbrt_camera_t* cam, const rbrt_render_opts_t* o,
                      TraceParams& P) {
    std::memset(&P, 0, sizeof(P));
    if (cam) P.cam = *cam;
    P.min_dist = o->min_dist;
    P.max_dist = o->max_dist;
    P.eps_frac = 1.0f / o->min_dist;  // triangle.rs:146
    for (int k = 0; k < 3; ++k) P.bg[k] = o->bg[k];
    P.max_depth = o->max_depth;
    P.seed_key = host_splitmix64(o->seed);
    P.n_spheres = s->n_spheres;
    P.n_meshes = s->n_meshes;
    P.n_elem_tris = s->n_elem_tris;
    P.spheres = s->d_spheres;
    P.elem_tris = s->d_elem_tris;
    P.elems = s->d_elems;
    P.materials = s->d_materials;
    P.meshes = s->d_meshes;
    P.tris = s->d_tris;
    P.counters = s->d_counters;
    return RBRT_OK;
}


// Scheduling only (never affects the image). A launch ends with every wave finishing the paths it still
// holds; that costs least when the last tiles handed out are cheap ones (primary rays that hit nothing end
// after one ray). Tiles go out in row-major order; this looks at the first and the last eighth of the rank's
// tiles (the size of one work shard) and says whether the EMPTY end is the first one, in which case the kernel
// walks the tiles backwards. One primary ray per tile centre against the objects' bounding boxes.
bool empty_end_is_first(const rbrt_hip_scene* s, const rbrt_camera_t& cam, uint32_t tiles_x, uint32_t rank, uint32_t world,
                        uint32_t n_local) {
    const auto tile_hits = [&](uint32_t tl) {
        const uint32_t tile = tl * world + rank;
        uint32_t ty, tx;
        tile_xy(tile, tiles_x, ty, tx);
        const double col = tx * RBRT_TILE + 0.5 * RBRT_TILE, row = ty * RBRT_TILE + 0.5 * RBRT_TILE;
        const double col_mm = (col - double(cam.img_width_pix / 2)) * cam.mm_per_pix_hor;
        const double row_mm = (row - double(cam.img_height_pix / 2)) * cam.mm_per_pix_vert;
        double d[3];
        for (int c = 0; c < 3; ++c)
            d[c] = (cam.img_center_point[c] + 0.001 * col_mm * cam.right[c] - 0.001 * row_mm * cam.up[c]) - cam.position[c];
        for (const auto& b : s->h_bounds) {
            double tn = 0.0, tf = 1e300;
            for (int c = 0; c < 3; ++c) {
                const double inv = 1.0 / d[c];
                double t0 = (b.lo[c] - cam.position[c]) * inv, t1 = (b.hi[c] - cam.position[c]) * inv;
                if (t0 > t1) std::swap(t0, t1);
                if (t0 == t0 && t0 > tn) tn = t0;
                if (t1 == t1 && t1 < tf) tf = t1;
            }
            if (tn <= tf) return true;
        }
        return false;
    };
    const uint32_t n_probe = std::max<uint32_t>(1, n_local / kWorkShards);
    uint32_t first = 0, last = 0;
    for (uint32_t i = 0; i < n_probe; ++i) first += tile_hits(i) ? 1u : 0u, last += tile_hits(n_local - 1u - i) ? 1u : 0u;
    return first < last;
}

// What scene_create is told about the call it serves (the ABI's rbrt_hip_scene_create knows nothing: a handle).
struct CreateHint {
    bool one_shot = false;       // rbrt_hip_render: one render, then the scene is destroyed
    double render_s_est = 0.0;   // ... and about this long (0: unknown)
};

// Which builder makes a mesh's FIRST tree: the one that costs the call less. Measured on an MI355X box (tools/create_sweep.py,
// profiles/r05_create_sweep.txt; the table below); frames through the device's tree take kDeviceTreeSlowdown longer
// (2.5-3.5 % measured). A handle starts with the cheaper build and, where that was the device's, gets the host's tree
// in the background (struct Refine); a one-shot call adds what the slower frames of ITS render would cost.
//   entries        500   2000   8000   32000   69451   262144   871414
//   host, ms      1.9    8.0    6.6    18.7    31.7    104.7    310.0    (build; spatial splits; threaded from 4096 entries, 16 threads)
//   device, ms    0.90   1.07   1.31    1.92    2.44     3.67     5.46   (upload + build: a dozen dependent stages, ~25 rounds in 4 batches)
constexpr double kHostBuildSecPerTri = 4.0e-6, kHostThreadedFrom = 4096.0, kDeviceBuildSec0 = 0.9e-3, kDeviceBuildSecPerDoubling = 0.18e-3,
                 kDeviceBuildSecPerTri = 3.0e-9, kDeviceTreeSlowdown = 0.04;
bool device_builder_is_cheaper(uint32_t n_total, const CreateHint& hint) {
    const double n = double(n_total);
    const double threads = double(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
    // (the top of the host's tree is built by one thread)
    const double host_s = kHostBuildSecPerTri * n * (n < kHostThreadedFrom ? 1.0 : 0.06 + 0.9 / threads);
    const double device_s = kDeviceBuildSec0 + kDeviceBuildSecPerDoubling * std::log2(std::max(n, 500.0) / 500.0) + kDeviceBuildSecPerTri * n +
                            (hint.one_shot ? kDeviceTreeSlowdown * hint.render_s_est : 0.0);
    return n_total >= 8u && device_s < host_s;
}

int scene_create_impl(const rbrt_scene_t* scene, int device, rbrt_hip_scene_t** out, const CreateHint& hint);

}  // namespace

extern "C" {

int rbrt_hip_abi_version(void) { return RBRT_ABI_VERSION; }

const char* rbrt_hip_last_error(void) { return g_last_error.c_str(); }

void rbrt_render_opts_default(rbrt_render_opts_t* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->spp = 5;             // src/main.rs:47
    o->max_depth = 50;      // lib.rs:99
    o->min_dist = 0.001f;   // lib.rs:44
    o->max_dist = 2000.0f;  // lib.rs:45
    o->bg[0] = 0.05f, o->bg[1] = 0.05f, o->bg[2] = 0.8f;  // lib.rs:89-93
    o->seed = 1;
    o->tile_rank = 0;
    o->tile_world = 1;
    o->flags = RBRT_FLAG_NONE;
}

int rbrt_hip_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return 0;
    }
    return n;
}

size_t rbrt_hip_packed_pixels(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_world) {
    const uint32_t tx = (width + RBRT_TILE - 1) / RBRT_TILE, ty = (height + RBRT_TILE - 1) / RBRT_TILE;
    const uint32_t world = tile_world ? tile_world : 1;
    return size_t(local_tiles_of(tx * ty, tile_rank, world)) * 64u;
}

void rbrt_hip_tile_xy(uint32_t tile, uint32_t tiles_x, uint32_t* tile_row, uint32_t* tile_col) {
    uint32_t ty = 0, tx = 0;
    if (tiles_x) tile_xy(tile, tiles_x, ty, tx);
    if (tile_row) *tile_row = ty;
    if (tile_col) *tile_col = tx;
}
uint32_t rbrt_hip_tile_number(uint32_t tile_row, uint32_t tile_col, uint32_t tiles_x) {
    return tiles_x ? tile_number(tile_row, tile_col, tiles_x) : 0u;
}

int rbrt_hip_scene_create(const rbrt_scene_t* scene, int device, rbrt_hip_scene_t** out) {
    return scene_create_impl(scene, device, out, CreateHint());
}

}  // extern "C"

namespace {

int scene_create_impl(const rbrt_scene_t* scene, int device, rbrt_hip_scene_t** out, const CreateHint& hint) {
    if (!scene || !out) return fail(RBRT_ERR_INVALID_ARG, "scene_create: null argument");
    *out = nullptr;
    if (scene->n_spheres && !scene->spheres) return fail(RBRT_ERR_INVALID_ARG, "spheres is null");
    if (scene->n_meshes && !scene->meshes) return fail(RBRT_ERR_INVALID_ARG, "meshes is null");
    if (scene->n_triangles && !scene->triangles) return fail(RBRT_ERR_INVALID_ARG, "triangles is null");
    if (uint64_t(scene->n_spheres) + scene->n_triangles + scene->n_meshes > uint64_t(kMaxObjects))
        return fail(RBRT_ERR_UNSUPPORTED, "more than 255 objects (spheres + triangles + meshes) in one scene");
    for (uint32_t i = 0; i < scene->n_spheres; ++i)
        if (int rc = check_material(scene->spheres[i].mat)) return rc;
    for (uint32_t i = 0; i < scene->n_triangles; ++i)
        if (int rc = check_material(scene->triangles[i].mat)) return rc;
    // Scene::elements order: the given one (every sphere and triangle exactly once), else spheres then triangles
    const uint32_t n_elem = scene->n_spheres + scene->n_triangles;
    std::vector<uint32_t> elems(n_elem);
    for (uint32_t e = 0; e < n_elem; ++e) elems[e] = e < scene->n_spheres ? e : (0x80000000u | (e - scene->n_spheres));
    if (scene->element_order && n_elem) {
        std::vector<uint8_t> seen_s(scene->n_spheres, 0), seen_t(scene->n_triangles, 0);
        for (uint32_t e = 0; e < n_elem; ++e) {
            const uint32_t d = scene->element_order[e], idx = d & 0x7FFFFFFFu;
            std::vector<uint8_t>& seen = (d >> 31) ? seen_t : seen_s;
            if (idx >= seen.size() || seen[idx]) return fail(RBRT_ERR_INVALID_ARG, "element_order must name every sphere and triangle exactly once");
            seen[idx] = 1;
            elems[e] = d;
        }
    }
    for (uint32_t i = 0; i < scene->n_meshes; ++i) {
        const rbrt_mesh_t& m = scene->meshes[i];
        if (int rc = check_material(m.mat)) return rc;
        // The BVH is at most kMaxBvhDepth + 1 inner levels deep with <= kLeafMax triangles per leaf: 4 << 21 =
        // 8,388,608 triangles is what always fits, whatever their arrangement (bvh.cpp capacity()).
        if (m.n_total > (4u << (kMaxBvhDepth + 1)))
            return fail(RBRT_ERR_UNSUPPORTED, "mesh has more than 8,388,608 triangles (BVH depth budget)");
        if (m.n_total && (!m.v0x || !m.v0y || !m.v0z || !m.e1x || !m.e1y || !m.e1z || !m.e2x || !m.e2y ||
                          !m.e2z || !m.nx || !m.ny || !m.nz || !m.is_padding))
            return fail(RBRT_ERR_INVALID_ARG, "mesh array pointer is null");
    }
    const double t_create0 = now_s();
    if (int rc = ensure_device(device)) return rc;

    rbrt_hip_scene* s = new rbrt_hip_scene();
    s->device = device;
    s->create_times.hip_init_s = now_s() - t_create0;
    double t_upload = 0.0, t_build = 0.0;
    s->n_spheres = scene->n_spheres;
    s->n_meshes = scene->n_meshes;
    s->n_elem_tris = scene->n_triangles;
    auto bail = [&](int rc) {
        rbrt_hip_scene_destroy(s);
        return rc;
    };
#define HIP_TRY_BAIL(expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return bail(fail(_e == hipErrorOutOfMemory ? RBRT_ERR_OOM : RBRT_ERR_HIP,                   \
                             std::string(#expr) + ": " + hipGetErrorString(_e)));                       \
    } while (0)

    for (uint32_t i = 0; i < scene->n_spheres; ++i) {
        rbrt_hip_scene::Bound b;
        for (int c = 0; c < 3; ++c)
            b.lo[c] = scene->spheres[i].center[c] - scene->spheres[i].radius, b.hi[c] = scene->spheres[i].center[c] + scene->spheres[i].radius;
        s->h_bounds.push_back(b);
    }
    for (uint32_t i = 0; i < scene->n_triangles; ++i) {
        rbrt_hip_scene::Bound b;
        for (int c = 0; c < 3; ++c) {
            const float x0 = scene->triangles[i].corners[0][c], x1 = scene->triangles[i].corners[1][c], x2 = scene->triangles[i].corners[2][c];
            b.lo[c] = std::min(x0, std::min(x1, x2)), b.hi[c] = std::max(x0, std::max(x1, x2));
        }
        s->h_bounds.push_back(b);
    }
    for (uint32_t i = 0; i < scene->n_meshes; ++i) {
        rbrt_hip_scene::Bound b;
        for (int c = 0; c < 3; ++c) b.lo[c] = scene->meshes[i].bbox_lo[c], b.hi[c] = scene->meshes[i].bbox_hi[c];
        s->h_bounds.push_back(b);
    }
    std::vector<DevSphere> spheres(scene->n_spheres);
    std::vector<DevTriangle> etris(scene->n_triangles);
    std::vector<DevMaterial> mats(n_elem + scene->n_meshes);  // [object id]: elements in their order, then meshes
    auto put_mat = [&](size_t k, const rbrt_material_t& m) {
        for (int c = 0; c < 3; ++c) mats[k].albedo[c] = m.albedo[c];
        mats[k].param = m.param;
        mats[k].kind = m.kind;
    };
    // A scene WITHOUT triangle elements is tested by the kernels' sphere-only loop, where element e IS device sphere e (no
    // element table is uploaded): the device array is therefore laid out in Scene::elements order, so that a permuted
    // `element_order` gives every sphere its own material, object id and place in a tie (scene.rs:23-31). With triangle
    // elements the table `elems` names the spheres by their position in the caller's array, which is kept.
    for (uint32_t i = 0; i < scene->n_spheres; ++i) {
        const uint32_t src = scene->n_triangles == 0 ? elems[i] : i;
        for (int c = 0; c < 3; ++c) spheres[i].center[c] = scene->spheres[src].center[c];
        spheres[i].radius = scene->spheres[src].radius;
    }
    for (uint32_t i = 0; i < scene->n_triangles; ++i) {  // BasicTriangle::new (triangle.rs:19-28): edges and normal, f32, unfused
        const rbrt_triangle_t& t = scene->triangles[i];
        DevTriangle& o = etris[i];
        for (int c = 0; c < 3; ++c) o.v0[c] = t.corners[0][c], o.e0[c] = t.corners[1][c] - t.corners[0][c], o.e1[c] = t.corners[2][c] - t.corners[0][c];
        const float cx = o.e0[1] * o.e1[2] - o.e0[2] * o.e1[1], cy = o.e0[2] * o.e1[0] - o.e0[0] * o.e1[2], cz = o.e0[0] * o.e1[1] - o.e0[1] * o.e1[0];
        const float len = std::sqrt((cx * cx + cy * cy) + cz * cz);  // vec3.rs:111-126: length, then three divisions
        o.normal[0] = cx / len, o.normal[1] = cy / len, o.normal[2] = cz / len;
    }
    for (uint32_t e = 0; e < n_elem; ++e)
        put_mat(e, (elems[e] >> 31) ? scene->triangles[elems[e] & 0x7FFFFFFFu].mat : scene->spheres[elems[e]].mat);
    std::vector<DevMesh> meshes(scene->n_meshes);
    // One triangle array for the scene (leaf links are absolute positions in it): mesh i owns the records
    // [tri_base[i], tri_base[i] + cap[i]), cap = what its builder can emit at most (bvh.h bvh_record_capacity: the scan-visible
    // entries, the host builder's duplicated references, a dummy).
    std::vector<uint32_t> tri_base(scene->n_meshes, 0);
    uint64_t tri_total = 0;
    for (uint32_t i = 0; i < scene->n_meshes; ++i) {
        tri_base[i] = uint32_t(tri_total);
        tri_total += bvh_record_capacity(scene->meshes[i].n_total);
        if (tri_total >= (1ull << 25)) return bail(fail(RBRT_ERR_UNSUPPORTED, "more than 2^25 triangle records in one scene"));
    }
    {
        void* p = nullptr;
        const size_t bytes = std::max<size_t>(size_t(tri_total) * sizeof(BvhTri), 64);
        HIP_TRY_BAIL(dev_alloc(s, bytes, &p));
        s->d_tris = static_cast<BvhTri*>(p);  // (records no leaf points at are never read: left as they are)
    }
    // Builder of each mesh's first tree: whichever costs this call less (device_builder_is_cheaper), the host's tree
    // following in the background for a handle (struct Refine); RBRT_BVH_BUILDER = host | device forces one and nothing
    // follows. The device builder declines what it does not handle (tiny meshes, a tree deeper than the traversal stack
    // allows) and the host builder takes over.
    int builder_mode = 0;  // 0 by cost, 1 host, 2 device
    if (const char* e = std::getenv("RBRT_BVH_BUILDER")) builder_mode = !std::strcmp(e, "host") ? 1 : !std::strcmp(e, "device") ? 2 : 0;
    bool refine_allowed = builder_mode == 0 && !hint.one_shot;
    s->one_shot = hint.one_shot;
    if (const char* e = std::getenv("RBRT_BVH_REFINE")) refine_allowed = refine_allowed && e[0] != '0';
    uint32_t device_min_tris = 0;  // (lab: a threshold on the entry count instead of the cost rule)
    bool device_min_set = false;
    {
        std::string err;
        if (!lab_u32("RBRT_BVH_DEVICE_MIN", 0, 1ll << 30, device_min_tris, err)) return bail(fail(RBRT_ERR_INVALID_ARG, err));
        device_min_set = lab_env("RBRT_BVH_DEVICE_MIN") != nullptr;
    }
    std::vector<uint8_t> device_built(scene->n_meshes, 0);
    std::vector<uint32_t> n_valid_of(scene->n_meshes, 0);
    const double t_meshes0 = now_s();
    for (uint32_t i = 0; i < scene->n_meshes; ++i) {
        const rbrt_mesh_t& m = scene->meshes[i];
        put_mat(n_elem + i, m.mat);
        DevMesh& dm = meshes[i];
        Normal4* d_normals = nullptr;
        {
            void* p = nullptr;
            HIP_TRY_BAIL(dev_alloc(s, std::max<size_t>(size_t(m.n_total) * sizeof(Normal4), 16), &p));
            d_normals = static_cast<Normal4*>(p);
        }
        bool built = false;
        const bool try_device = builder_mode == 2 || (builder_mode == 0 && (device_min_set ? m.n_total >= device_min_tris
                                                                                           : device_builder_is_cheaper(m.n_total, hint)));
        if (try_device && m.n_total >= 8) {
            DeviceBvhResult r;
            const double tb0 = now_s();
            double up = 0.0;
            const hipError_t e = device_build_mesh(m, s->d_tris + tri_base[i], tri_base[i], d_normals, &r, &up);
            t_upload += up, t_build += now_s() - tb0 - up;
            if (e != hipSuccess)
                return bail(fail(e == hipErrorOutOfMemory ? RBRT_ERR_OOM : RBRT_ERR_HIP, std::string("device BVH build: ") + hipGetErrorString(e)));
            if (r.ok) {
                s->allocs.push_back(r.d_nodes);
                dm.nodes = r.d_nodes;
                dm.max_e12 = r.max_e12;
                dm.n_nodes = r.n_nodes;
                dm.n_tris = r.n_valid;
                s->stack_need = std::max(s->stack_need, 3u * (r.max_depth + 1u) + 1u);
                s->total_nodes += r.n_nodes;
                s->total_tris += r.n_valid;
                s->n_device_built += 1;
                device_built[i] = 1, n_valid_of[i] = r.n_valid;
                built = true;
            }
        }
        if (!built) {
            const double tb0 = now_s();
            BvhBuildResult bvh = build_bvh(m);
            if (tri_base[i] != 0)
                for (BvhNode4& nd : bvh.nodes)
                    for (int c = 0; c < 4; ++c)
                        if (nd.child[c] < 0 && nd.child[c] != kNoChild) {
                            const uint32_t leaf = uint32_t(~nd.child[c]);
                            nd.child[c] = ~int32_t((((leaf >> kLeafBits) + tri_base[i]) << kLeafBits) | (leaf & uint32_t(kLeafMax - 1)));
                        }
            std::vector<Normal4> normals(m.n_total);
            for (uint32_t k = 0; k < m.n_total; ++k) normals[k] = Normal4{m.nx[k], m.ny[k], m.nz[k], 0.0f};
            const double tb1 = now_s();
            BvhNode4* d_nodes = nullptr;
            if (int rc = upload(s, bvh.nodes, &d_nodes)) return bail(rc);
            if (!bvh.tris.empty())
                HIP_TRY_BAIL(hipMemcpy(s->d_tris + tri_base[i], bvh.tris.data(), bvh.tris.size() * sizeof(BvhTri), hipMemcpyHostToDevice));
            if (!normals.empty())
                HIP_TRY_BAIL(hipMemcpy(d_normals, normals.data(), normals.size() * sizeof(Normal4), hipMemcpyHostToDevice));
            t_build += tb1 - tb0, t_upload += now_s() - tb1;
            dm.nodes = d_nodes;
            dm.max_e12 = bvh.max_e12;
            dm.n_nodes = uint32_t(bvh.nodes.size());
            dm.n_tris = uint32_t(bvh.tris.size());
            s->stack_need = std::max(s->stack_need, bvh.stack_need);
            s->total_nodes += bvh.nodes.size();
            s->total_tris += bvh.tris.size();
        }
        dm.tris = s->d_tris, dm.normals = d_normals;
        float diag2 = 0.0f;
        for (int c = 0; c < 3; ++c) {
            dm.bbox_lo[c] = m.bbox_lo[c];
            dm.bbox_hi[c] = m.bbox_hi[c];
            dm.center[c] = 0.5f * m.bbox_lo[c] + 0.5f * m.bbox_hi[c];
            float h = 0.5f * m.bbox_hi[c] - 0.5f * m.bbox_lo[c];
            diag2 += h * h;
        }
        // Radius bounds |v - center| for every indexed vertex; padded a little for its own rounding.
        dm.radius = std::sqrt(diag2) * 1.0001f;
        if (!std::isfinite(dm.radius)) dm.radius = std::numeric_limits<float>::max();
    }
    const double t_meshes1 = now_s();
    if (int rc = upload(s, spheres, &s->d_spheres)) return bail(rc);
    if (scene->n_triangles != 0) {
        if (int rc = upload(s, etris, &s->d_elem_tris)) return bail(rc);
        if (int rc = upload(s, elems, &s->d_elems)) return bail(rc);
    }
    if (int rc = upload(s, mats, &s->d_materials)) return bail(rc);
    if (int rc = upload(s, meshes, &s->d_meshes)) return bail(rc);
    {
        std::vector<DevCounters> z(1);
        std::memset(z.data(), 0, sizeof(DevCounters));
        if (int rc = upload(s, z, &s->d_counters)) return bail(rc);
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail(fail(RBRT_ERR_HIP, "hipGetDeviceProperties failed"));
        int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        std::string err;
        uint32_t waves_per_cu = 0, poison = 0, stripes = s->work_stripes, stripes_overlap = s->work_stripes_overlap;
        const bool knobs_ok =
            lab_u32("RBRT_POOL", 128, 256, s->pool, err, {128, 256}) && lab_u32("RBRT_LDS_STACK", 1, kStackMax, s->stack_entries, err) &&
            lab_u32("RBRT_LEAF_ROUND", 1, 64, s->leaf_round, err) && lab_u32("RBRT_Y_HIGH", 1, 64, s->y_high_water, err) &&
            lab_u32("RBRT_Y_HIGH_PARKED", 1, 256, s->y_high_min_parked, err) && lab_u32("RBRT_Y_LOW", 1, 64, s->y_low_water, err) &&
            lab_u32("RBRT_WAVES_PER_CU", 1, 32, waves_per_cu, err) && lab_u32("RBRT_LEAF_LEAVES", 1, 128, s->leaf_leaves, err) &&
            lab_u32("RBRT_SHARE_IDLE", 0, 64, s->share_idle, err) && lab_u32("RBRT_WORK_STRIPES", 0, 65536, stripes, err) &&
            lab_u32("RBRT_WORK_STRIPES_OVERLAP", 0, 65536, stripes_overlap, err) && lab_u32("RBRT_DRAIN_MODE", 0, 11, s->drain_mode, err) &&
            lab_u32("RBRT_SHADE_ROUNDS", 1, kMaxShadeRounds, s->shade_rounds, err) &&
            lab_u32("RBRT_SHADE_CONT_MIN", 1, 64, s->shade_cont_min, err) && lab_u32("RBRT_PIPELINE", 0, kMaxPipeline, s->pipeline, err) &&
            lab_u32("RBRT_POISON_SAMPLES", 0, 1, poison, err) && lab_u32("RBRT_PRIMARY_CULL", 0, 1, s->primary_cull, err) &&
            lab_u32("RBRT_TILE_ORDER", 0, 2, s->tile_order, err) && lab_u32("RBRT_TILE_CLASSES", 0, 4, s->tile_classes, err) &&
            lab_u32("RBRT_OVERLAP_WAVES_PER_CU", 0, 16, s->overlap_waves_per_cu, err) &&
            lab_u32("RBRT_TILE_ISOLATED_MODE", 0, 4, s->isolated_list_mode, err) && lab_u32("RBRT_TILE_TAIL_DIV", 1, 1024, s->tile_tail_div, err) &&
            lab_u32("RBRT_HELPERS", 0, 2, s->helpers_mode, err) && lab_u32("RBRT_HELPER_MIN_ITEMS", 1, 1 << 24, s->helper_min_items, err) &&
            lab_u32("RBRT_HELPER_MIN_LAUNCH_MI", 0, 4096, s->helper_min_launch_mi, err) && lab_u32("RBRT_HELPER_MIN_FREE", 1, 16, s->helper_min_free_per_cu, err) &&
            lab_u32("RBRT_HELPER_ROUNDS", 1, 16, s->helper_rounds, err);
        if (!knobs_ok) return bail(fail(RBRT_ERR_INVALID_ARG, err));
        s->tile_classes_set = lab_env("RBRT_TILE_CLASSES") != nullptr;
        s->trace_launches = lab_env("RBRT_TRACE_LAUNCHES") != nullptr;
        if ((stripes & (stripes - 1u)) != 0u || (stripes_overlap != kStripesAuto && (stripes_overlap & (stripes_overlap - 1u)) != 0u))  // the kernel shifts
            return bail(fail(RBRT_ERR_INVALID_ARG, "lab knob RBRT_WORK_STRIPES / RBRT_WORK_STRIPES_OVERLAP must be 0 or a power of two"));
        s->work_stripes = stripes, s->work_stripes_overlap = stripes_overlap;
        s->drain_mode &= 11u;
        s->poison_samples = poison != 0;
        if (const char* e = lab_env("RBRT_SHARE_BELOW")) s->share_below = std::strtoull(e, nullptr, 10);
        if (s->stack_entries > s->stack_need) s->stack_entries = s->stack_need;
        // resident waves per CU: LDS-limited (160 KiB per CU), at most 5 per SIMD (VGPR budget)
        int per_cu = int((160u * 1024u) / megakernel_lds_bytes(s->pool, s->stack_entries, s->n_spheres, s->n_meshes, s->n_elem_tris));
        if (per_cu > 20) per_cu = 20;
        if (per_cu < 1) per_cu = 1;
        if (waves_per_cu != 0) per_cu = int(waves_per_cu), s->waves_fixed = true;
        s->n_waves = uint32_t(cus * per_cu);
        s->n_cus = uint32_t(cus);
        s->hw_queues = g_hw_queues_at_load;
        // per-wave scratch is indexed by workgroup (= wave): no grid is larger than n_waves; two more waves per CU for helper
        // launches behind a full grid (a blocking caller's launch under RBRT_HELPERS=2)
        s->scratch_waves = s->n_waves + 2u * s->n_cus;
        // lanes up front: rbrt_hip_render_device then never allocates lanes (which synchronises the device); a one-shot
        // call makes the lanes its batches will use (rbrt_hip_render)
        const double t_lanes0 = now_s();
        if (lab_env("RBRT_TRACE_CREATE"))
            std::fprintf(stderr, "[rbrt_hip] scene_create: device %.3f ms, record array %.3f, meshes %.3f (upload %.3f, build %.3f), tables + properties %.3f\n",
                         s->create_times.hip_init_s * 1e3, (t_meshes0 - t_create0 - s->create_times.hip_init_s) * 1e3, (t_meshes1 - t_meshes0) * 1e3,
                         t_upload * 1e3, t_build * 1e3, (t_lanes0 - t_meshes1) * 1e3);
        if (int rc = ensure_lanes(s, std::max(hint.one_shot ? 1u : kLanesAtCreate, s->pipeline))) return bail(rc);
        // (the library's device code, loaded now instead of inside the first render)
        if (launch_code_load(nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            return bail(fail(RBRT_ERR_HIP, "the library's device code does not load on this GPU (built for gfx950)"));
        s->create_times.lanes_s = now_s() - t_lanes0;
    }
    // the host builder's tree of the device-built meshes, in the background (struct Refine)
    if (refine_allowed && s->n_device_built != 0) {
        std::unique_ptr<Refine> r(new Refine());
        r->device = device;
        r->d_tris = s->d_tris, r->tri_total = size_t(tri_total);
        r->meshes = meshes, r->tri_base = tri_base, r->n_valid = n_valid_of, r->wanted = device_built;
        const unsigned hc = std::thread::hardware_concurrency();
        r->max_threads = std::max(1u, hc / 2u);  // (the caller's thread goes on issuing launches)
        Refine* rp = r.get();
        try {
            r->th = std::thread([rp]() {
                try {
                    rp->run();
                } catch (...) {  // (no memory for the copy of the records or for the tree: the device's trees stay)
                    for (void* p : rp->allocs) (void)hipFree(p);
                    rp->allocs.clear();
                    rp->error = "out of host memory";
                    rp->state.store(2, std::memory_order_release);
                }
            });
            s->refine = std::move(r);
        } catch (...) {  // (no thread to be had: the device's tree stays)
        }
    }
    s->create_times.upload_s = t_upload, s->create_times.bvh_build_s = t_build;
    s->create_times.meshes_device_built = s->n_device_built, s->create_times.meshes_host_built = s->n_meshes - s->n_device_built;
    s->create_times.create_s = now_s() - t_create0;
    *out = s;
    return RBRT_OK;
}

}  // namespace

extern "C" {

int rbrt_hip_scene_destroy(rbrt_hip_scene_t* s) {
    if (!s) return RBRT_OK;
    (void)hipSetDevice(s->device);
    if (s->watcher.joinable()) {  // (before anything it looks at goes away)
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->watcher_stop = true;
        }
        s->cv.notify_all();
        s->watcher.join();
    }
    if (Refine* r = s->refine.get()) {  // a background build nobody will use: cancelled, its allocations released
        r->cancel.store(true);
        if (r->th.joinable()) r->th.join();
        for (void* p : r->allocs) (void)hipFree(p);
        r->allocs.clear();
    }
    const bool trace_destroy = lab_env("RBRT_TRACE_CREATE");
    const double td0 = now_s();
    (void)hipDeviceSynchronize();  // lane streams included
    const double td1 = now_s();
    double t_big = 0.0;
    for (auto& L : s->lanes) {
        if (L.stream && L.stream_pool_key >= 0) give_pooled_stream(L.stream_pool_key, L.stream);  // (idle: the device was synchronised above)
        else if (L.stream) (void)hipStreamDestroy(L.stream);
        if (L.ev_helper) (void)hipEventDestroy(L.ev_helper);
        if (L.ev_carried) (void)hipEventDestroy(L.ev_carried);
        if (L.ev_ready) (void)hipEventDestroy(L.ev_ready);
        if (L.ev_traced) (void)hipEventDestroy(L.ev_traced);
        for (auto& B : L.bufs) {
            if (B.ev_resolved) (void)hipEventDestroy(B.ev_resolved);
            const double tb = now_s();
            if (B.d_sample_buf) (void)hipFree(B.d_sample_buf);
            t_big += now_s() - tb;
        }
        for (auto& T : L.tiles) {
            if (T.ev_lists) (void)hipEventDestroy(T.ev_lists);
            if (T.ev_free) (void)hipEventDestroy(T.ev_free);
        }
    }
    const double td2 = now_s();
    if (s->prep_stream) (void)hipStreamDestroy(s->prep_stream);
    if (s->aux_stream) (void)hipStreamDestroy(s->aux_stream);
    const double td3 = now_s();
    if (s->h_zeros) (void)hipHostFree(s->h_zeros);
    const double td4 = now_s();
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->d_acc) (void)hipFree(s->d_acc);
    const double td5 = now_s();
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    if (trace_destroy)
        std::fprintf(stderr, "[rbrt_hip] scene_destroy: sync %.3f ms, lanes %.3f (sample buffers %.3f), streams %.3f, pinned %.3f, %zu allocations %.3f, %zu events %.3f\n",
                     (td1 - td0) * 1e3, (td2 - td1) * 1e3, t_big * 1e3, (td3 - td2) * 1e3, (td4 - td3) * 1e3, s->allocs.size(), (td5 - td4) * 1e3,
                     s->events.size(), (now_s() - td5) * 1e3);
    delete s;
    return RBRT_OK;
}

int rbrt_hip_scene_set_timing(rbrt_hip_scene_t* s, int enable) {
    if (!s) return fail(RBRT_ERR_INVALID_ARG, "null scene");
    std::lock_guard<std::mutex> watcher_lock(s->mu);
    s->timing = enable != 0;
    s->timing_overflow = false;
    s->events_used = 0;
    s->n_full_grid = s->n_half_grid = 0;
    s->n_helper_launches = 0;
    return RBRT_OK;
}

// How the launches since set_timing(1) were issued: the grid of a launch depends on whether another one was still
// running when it was issued (grid_for), i.e. on host timing; A/B sweeps read the mix beside the times.
int rbrt_hip_scene_launch_mix(rbrt_hip_scene_t* s, uint32_t* n_full_grid, uint32_t* n_half_grid) {
    if (!s) return fail(RBRT_ERR_INVALID_ARG, "null scene");
    if (n_full_grid) *n_full_grid = s->n_full_grid;
    if (n_half_grid) *n_half_grid = s->n_half_grid;
    return RBRT_OK;
}

// Helper launches (api.cpp "Elastic launches") issued since set_timing(1).
int rbrt_hip_scene_helper_launches(rbrt_hip_scene_t* s, uint32_t* n) {
    if (!s || !n) return fail(RBRT_ERR_INVALID_ARG, "helper_launches: null argument");
    std::lock_guard<std::mutex> watcher_lock(s->mu);
    *n = s->n_helper_launches;
    return RBRT_OK;
}

int rbrt_hip_scene_kernel_ms(rbrt_hip_scene_t* s, float* trace_ms, float* resolve_ms, uint32_t* n_launches) {
    if (!s) return fail(RBRT_ERR_INVALID_ARG, "null scene");
    if (!s->timing) return fail(RBRT_ERR_INVALID_ARG, "timing is not enabled on this scene");
    HIP_TRY(hipSetDevice(s->device));
    float tsum = 0.0f, rsum = 0.0f;
    uint32_t n = 0;
    for (size_t b = 0; b + 3 <= s->events_used; b += 3) {
        HIP_TRY(hipEventSynchronize(s->events[b + 2]));
        float t = 0.0f, r = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, s->events[b], s->events[b + 1]));
        HIP_TRY(hipEventElapsedTime(&r, s->events[b + 1], s->events[b + 2]));
        tsum += t;
        rsum += r;
        ++n;
    }
    if (trace_ms) *trace_ms = tsum;
    if (resolve_ms) *resolve_ms = rsum;
    if (n_launches) *n_launches = n;
    return RBRT_OK;
}

// Samples [s_begin, s_end) of a render of o->spp samples per pixel. `acc` holds (and receives) the per-pixel running
// sums in sample order -- lib.rs:95-100's `color +=` -- packed like the radiance output; it is read unless
// s_begin == 0 and written unless s_end == o->spp, in which case the mean (lib.rs:101) and the quantisation go out.
static int render_samples(rbrt_hip_scene_t* s, const rbrt_camera_t* cam, const rbrt_render_opts_t* o, void* stream_v,
                   uint32_t s_begin, uint32_t s_end, float* acc, float* d_radiance, uint8_t* d_rgb8) {
    if (!s || !cam || !o) return fail(RBRT_ERR_INVALID_ARG, "render: null argument");
    if (o->spp == 0) return fail(RBRT_ERR_INVALID_ARG, "spp must be >= 1");
    if (s_begin >= s_end || s_end > o->spp) return fail(RBRT_ERR_INVALID_ARG, "sample range must satisfy begin < end <= spp");
    if (o->max_depth > uint32_t(kMaxPathDepth))
        return fail(RBRT_ERR_UNSUPPORTED, "max_depth above the kernel limit of 64");
    if (cam->img_width_pix == 0 || cam->img_height_pix == 0)
        return fail(RBRT_ERR_INVALID_ARG, "image has zero pixels");
    if (uint64_t(cam->img_width_pix) * cam->img_height_pix >= (1ull << 32))
        return fail(RBRT_ERR_UNSUPPORTED, "image has 2^32 or more pixels");
    const uint32_t world = o->tile_world ? o->tile_world : 1;
    if (o->tile_rank >= world) return fail(RBRT_ERR_INVALID_ARG, "tile_rank >= tile_world");
    HIP_TRY(hipSetDevice(s->device));
    std::lock_guard<std::mutex> watcher_lock(s->mu);  // (the watcher of elastic launches looks at the lanes between calls, not during one)
    adopt_refined(s);  // (the background thread's trees, once they are on the device: this call's launches use them)
    hipStream_t stream = static_cast<hipStream_t>(stream_v);

    const uint32_t W = cam->img_width_pix, H = cam->img_height_pix;
    const uint32_t tiles_x = (W + RBRT_TILE - 1) / RBRT_TILE, tiles_y = (H + RBRT_TILE - 1) / RBRT_TILE;
    const uint32_t n_tiles = tiles_x * tiles_y;
    const uint32_t n_local = local_tiles_of(n_tiles, o->tile_rank, world);
    const size_t npix = size_t(n_local) * 64u;
    if (npix == 0) return RBRT_OK;

    // workspace: as many samples per batch as fit the cap (at least one)
    const size_t per_sample = npix * 3u * sizeof(float);
    size_t batch = workspace_cap_bytes() / per_sample;
    if (batch * npix > 0xFFF00000ull) batch = 0xFFF00000ull / npix;  // work items are 32-bit in the kernel
    if (batch < 1) batch = 1;
    if (batch > s_end - s_begin) batch = s_end - s_begin;
    const size_t need = batch * per_sample;
    // The pipeline is for STREAMS of launches. A call that finds the GPU idle and is not one of a stream -- the CLI's one
    // render, a blocking caller -- uses the lanes there are (four from scene_create: four more cost 20 ms to make, more than
    // thirteen batches of a 1920x1080x512 render gain from them), and only the lanes it launches on get their buffers: the first frame of a scene took 12.9 ms with three lanes' buffers to make
    // and 39.9 with eight lanes', against 5.0 with one (4.2 from the second frame on). The first call of a stream that is
    // issued into a busy GPU makes the rest, during the stream's first frames.
    const bool stats_call = (o->flags & RBRT_FLAG_COLLECT_STATS) != 0;
    const uint32_t depth_full = depth_for(s, need);
    const bool busy_at_call = other_launch_in_flight(s, nullptr);  // (asked once: every answer is up to eight hipEventQuery calls)
    const bool streams_now = !stats_call && depth_full > 1 && (s->streaming_hint || busy_at_call);
    const uint32_t depth = streams_now ? depth_full : std::min<uint32_t>(depth_full, uint32_t(std::max<size_t>(s->lanes.size(), 1)));
    if (int rc = ensure_lanes(s, depth)) return rc;
    const auto sync_lanes = [&]() -> int {  // everything in flight on the caller's stream and on the lanes
        HIP_TRY(hipStreamSynchronize(stream));
        for (auto& L : s->lanes) HIP_TRY(hipStreamSynchronize(L.stream));  // (helper launches are carried by lane streams)
        if (s->aux_stream) HIP_TRY(hipStreamSynchronize(s->aux_stream));
        return RBRT_OK;
    };
    // (running sums between the batches of one call: a call of one batch has none)
    if (!acc && batch < s_end - s_begin && per_sample > s->acc_bytes) {
        if (s->d_acc) {
            if (int rc = sync_lanes()) return rc;
            HIP_TRY(hipFree(s->d_acc));
            s->d_acc = nullptr, s->acc_bytes = 0;
        }
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, per_sample));
        s->d_acc = static_cast<float*>(p), s->acc_bytes = per_sample;
    }

    const bool stats = (o->flags & RBRT_FLAG_COLLECT_STATS) != 0;
    if (stats) {
        HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(DevCounters), stream));
        {   // the two atomicMin slots start at all-ones
            const unsigned long long ones = ~0ull;
            HIP_TRY(hipMemcpyAsync(&s->d_counters->diag[24], &ones, sizeof(ones), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(&s->d_counters->diag[28], &ones, sizeof(ones), hipMemcpyHostToDevice, stream));
        }
        s->stats_pending = true;
    }

    TraceParams P;
    fill_trace_params(s, cam, o, P);
    P.tiles_x = tiles_x, P.tiles_y = tiles_y, P.n_tiles = n_tiles;
    P.tiles_x_magic = div_magic_of(tiles_x);
    P.tiles_reversed = empty_end_is_first(s, *cam, tiles_x, o->tile_rank, world, n_local) ? 1u : 0u;
    if (s->tile_order != 0u) P.tiles_reversed = s->tile_order - 1u;  // (lab knob)
    const uint32_t rowmajor_reversed = P.tiles_reversed;  // (a launch whose list is in class order resets it; the next batch may not)
    P.tile_rank = o->tile_rank, P.tile_world = world, P.n_local_tiles = n_local;
    P.stack_entries = s->stack_entries;
    P.y_low_water = s->y_low_water;
    P.y_high_water = s->y_high_water < s->y_low_water ? s->y_low_water : s->y_high_water;
    P.y_high_min_parked = s->y_high_min_parked;
    P.leaf_round = s->leaf_round;
    P.leaf_leaves = s->leaf_leaves;
    P.share_idle = s->share_idle;
    P.drain_mode = s->drain_mode;
    P.work_stripes = s->work_stripes;  // (per launch: set where the launch's size is known)
    P.shade_rounds = s->shade_rounds;
    P.shade_cont_min = s->shade_cont_min;
    P.tile_tail_div = s->tile_tail_div;

    ResolveParams R;
    std::memset(&R, 0, sizeof(R));
    R.width = W, R.height = H, R.tiles_x = tiles_x, R.n_tiles = n_tiles;
    R.tile_rank = o->tile_rank, R.tile_world = world, R.n_local_tiles = n_local;
    R.inv_spp = 1.0f / float(o->spp);  // lib.rs:101
    R.acc = acc ? acc : s->d_acc;
    R.out_radiance = d_radiance;
    R.out_rgb8 = d_rgb8;

    const size_t n_batches = (size_t(s_end - s_begin) + batch - 1) / batch;
    s->last_batch = uint32_t(batch), s->last_n_batches = uint32_t(n_batches);
    if (s->timing && s->events_used + 3 * n_batches > kMaxTimedLaunches * 3) s->timing_overflow = true;
    const bool timing = s->timing && !s->timing_overflow;
    if (timing) {
        while (s->events.size() < s->events_used + 3 * n_batches) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            s->events.push_back(e);
        }
    }
    // For a stream every lane's buffers are sized here, at the first call that needs them, not when a lane first comes up in
    // the rotation (a hipMalloc in the middle of a stream of frames); otherwise a lane's are sized when a batch goes to it.
    const bool tile_pass = s->primary_cull != 0;
    const size_t lists_need = size_t(kTileListHeader) + 2u * size_t(n_local);
    const auto size_lane = [&](uint32_t li) -> int {
        rbrt_hip_scene::Lane& L = s->lanes[li];
        if (!L.d_gseq) {  // the lane's per-wave scratch: scatter records beyond the four in LDS (written before they are read: no
                          // initial value), the overflow of the LDS stacks
            void* p = nullptr;
            const size_t gseq_bytes = (megakernel_gseq_bytes(s->scratch_waves, s->pool) + kSlabAlign - 1) & ~(kSlabAlign - 1);
            HIP_TRY(dev_alloc(s, gseq_bytes + megakernel_gstack_bytes(s->scratch_waves), &p));
            L.d_gseq = static_cast<uint32_t*>(p);
            L.d_gstack = reinterpret_cast<uint32_t*>(static_cast<char*>(p) + gseq_bytes);
        }
        for (auto& B : L.bufs) {
            if (need <= B.sample_buf_bytes) continue;
            if (B.d_sample_buf) {
                if (int rc = sync_lanes()) return rc;
                HIP_TRY(hipFree(B.d_sample_buf));
                B.d_sample_buf = nullptr, B.sample_buf_bytes = 0;
            }
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, need));
            B.d_sample_buf = static_cast<float*>(p), B.sample_buf_bytes = need;
        }
        if (tile_pass && (n_tiles > L.tile_cull_words || lists_need > L.tile_lists_words)) {
            if (L.tiles[0].d_cull || L.tiles[0].d_lists) {
                if (int rc = sync_lanes()) return rc;
                if (s->prep_stream) HIP_TRY(hipStreamSynchronize(s->prep_stream));
            }
            for (auto& T : L.tiles) {
                // (a set that has grown leaves its old arrays in the slab: a few hundred KB per frame size, until destroy)
                T.d_cull = T.d_lists = nullptr;
                T.key_valid = false, T.free_recorded = false;
                void* p = nullptr;
                HIP_TRY(dev_alloc(s, size_t(n_tiles) * sizeof(uint32_t), &p));
                T.d_cull = static_cast<uint32_t*>(p);
                HIP_TRY(dev_alloc(s, lists_need * sizeof(uint32_t), &p));
                T.d_lists = static_cast<uint32_t*>(p);
                if (!T.ev_lists) HIP_TRY(hipEventCreateWithFlags(&T.ev_lists, hipEventDisableTiming));
                if (!T.ev_free) HIP_TRY(hipEventCreateWithFlags(&T.ev_free, hipEventDisableTiming));
            }
            L.tile_cull_words = n_tiles, L.tile_lists_words = lists_need;
        }
        return RBRT_OK;
    };
    if (streams_now)
        for (uint32_t li = 0; li < depth; ++li)
            if (int rc = size_lane(li)) return rc;
    const auto ensure_prep_stream = [&]() -> int {
        if (s->prep_stream) return RBRT_OK;
        int prio_low = 0, prio_high = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);  // (numerically: low >= high)
        HIP_TRY(hipStreamCreateWithPriority(&s->prep_stream, hipStreamNonBlocking, prio_high));
        return RBRT_OK;
    };
    // (a one-shot scene's launch has the GPU to itself and runs its tile pass on its own stream: it makes the prep stream
    // only if a second batch ever overlaps the first -- creating and destroying one costs the call 0.3 ms)
    if (tile_pass && !s->one_shot)
        if (int rc = ensure_prep_stream()) return rc;
    // The tile pass of a lane that has never had one runs now, for this camera, on the caller's stream (the lanes' buffers
    // were made just above; in the middle of a stream of frames a first pass would have to find room beside resident waves).
    if (tile_pass && streams_now) {
        TraceParams T;
        fill_trace_params(s, cam, o, T);
        T.tiles_x = tiles_x, T.tiles_y = tiles_y, T.n_tiles = n_tiles;
        T.tile_rank = o->tile_rank, T.tile_world = world, T.n_local_tiles = n_local;
        T.tile_list_mode = list_mode_for(s, true);  // (what the launches of a stream use; an isolated launch makes its own list)
        T.tile_tail_div = s->tile_tail_div;
        T.tiles_reversed = P.tiles_reversed;
        // (the one-wave form of the list kernel: `streams_now` means a launch of this scene is in flight or about to be, and a
        // 256-thread workgroup waits for four free wave slots on one CU beside resident trace waves -- up to 10 ms for 21 us
        // of work, profiles/r04_default_kernel_stats.csv -- with the resolve chain of the caller's stream queued behind it)
        T.tile_lists_wide = 0u;
        for (uint32_t li = 0; li < depth; ++li) {
            rbrt_hip_scene::Lane& L = s->lanes[li];
            if (L.tiles[0].key_valid || L.tiles[1].key_valid) continue;
            rbrt_hip_scene::Lane::TileSet& S = L.tiles[0];
            T.tile_cull = S.d_cull, T.tile_lists = S.d_lists;
            HIP_TRY(launch_primary_cull(T, stream));
            HIP_TRY(hipEventRecord(S.ev_lists, stream));
            std::memset(&S.key, 0, sizeof(S.key));
            S.key.cam = *cam, S.key.rank = o->tile_rank, S.key.world = world, S.key.list_mode = T.tile_list_mode, S.key.min_dist = o->min_dist;
            S.key_valid = true;
            // (the lane this call's first launch takes: if that launch finds the GPU idle it wants the isolated launch's list
            // as well -- made here too, it does not have to wait for the prep stream while the launches behind it pile up)
            const uint32_t iso_mode = list_mode_for(s, false);
            if (li == s->next_lane % depth && iso_mode != T.tile_list_mode) {
                rbrt_hip_scene::Lane::TileSet& S1 = L.tiles[1];
                TraceParams T1 = T;
                T1.tile_list_mode = iso_mode;
                T1.tile_cull = S1.d_cull, T1.tile_lists = S1.d_lists;
                HIP_TRY(launch_primary_cull(T1, stream));
                HIP_TRY(hipEventRecord(S1.ev_lists, stream));
                S1.key = S.key, S1.key.list_mode = iso_mode;
                S1.key_valid = true;
            }
        }
    }
    const size_t ev0 = s->events_used;
    bool call_streams = false;  // this call was made while an earlier one's launch was still running (decided at its first batch)
    uint32_t prev_lane = ~0u;
    for (size_t b = 0; b < n_batches; ++b) {
        const uint32_t base = s_begin + uint32_t(b * batch);
        const uint32_t nb = uint32_t(std::min<size_t>(batch, s_end - base));
        // counting launches run alone on lane 0: their counters are reset and read on the caller's stream
        // (the lanes take turns; but a launch that finds nothing of this scene in flight and is not one of a stream -- a
        // blocking caller's -- goes to a lane that already has the tile tables of its camera, if there is one: with eight
        // lanes taking turns a caller who renders one view again and again would otherwise pay the tile pass eight times)
        uint32_t lane_no = stats ? 0u : s->next_lane % depth;
        // (a call's later batches find its earlier ones in flight)
        if (!stats && depth > 1 && s->primary_cull != 0 && !s->streaming_hint && b == 0 && !busy_at_call) {
            rbrt_hip_scene::Lane::TileKey want;
            std::memset(&want, 0, sizeof(want));
            want.cam = *cam, want.rank = o->tile_rank, want.world = world, want.list_mode = list_mode_for(s, false), want.min_dist = o->min_dist;
            for (uint32_t li = 0; li < depth; ++li)
                for (const auto& C : s->lanes[li].tiles)
                    if (C.key_valid && std::memcmp(&want, &C.key, sizeof(want)) == 0) lane_no = li;
        } else if (!stats) {
            ++s->next_lane;
            if (b != 0 && depth > 1 && lane_no == prev_lane) lane_no = s->next_lane++ % depth;  // (beside the batch before, not behind it)
        }
        prev_lane = lane_no;
        if (int rc = size_lane(lane_no)) return rc;  // (nothing to do for a lane of a stream)
        rbrt_hip_scene::Lane& L = s->lanes[lane_no];
        const bool piped = depth > 1 && !stats;
        hipStream_t ts = piped ? L.stream : stream;  // the trace launch's stream
        rbrt_hip_scene::Lane::Buf& B = L.bufs[0];
        if (piped) {
            // the lane's sample buffer and counters are free once the resolve of its previous launch has run
            if (B.in_use) HIP_TRY(hipStreamWaitEvent(L.stream, B.ev_resolved, 0));
        } else if (depth > 1) {
            if (int rc = sync_lanes()) return rc;  // a counting launch: nothing else in flight
        }
        L.open.valid = false;  // (the launch the watcher knew on this lane is being followed by another: no stream has ended)
        if (L.helper_pending) {
            // a helper launch joined the lane's last launch: its last waves may have drawn from the counters after that
            // launch's resolve had reset them (api.cpp "Elastic launches") -- this launch starts behind the helper launch, on
            // counters zeroed again
            // (from pinned memory: an asynchronous copy out of pageable memory makes the issuing thread wait for the stream)
            constexpr size_t counter_bytes = sizeof(unsigned long long) * kWorkShards * kWorkCounterStride;
            if (!s->h_zeros) {
                HIP_TRY(hipHostMalloc(&s->h_zeros, counter_bytes, hipHostMallocDefault));
                std::memset(s->h_zeros, 0, counter_bytes);
            }
            HIP_TRY(hipStreamWaitEvent(ts, L.ev_helper, 0));
            HIP_TRY(hipMemcpyAsync(B.d_work_counter, s->h_zeros, counter_bytes, hipMemcpyHostToDevice, ts));
            L.helper_pending = false;
        }
        P.sample_base = base;
        P.batch = nb;
        P.batch_magic = div_magic_of(nb);
        P.n_items = uint64_t(npix) * nb;
        // (a caller that streams launches keeps doing so: the first launch after a pause -- the GPU is idle, but the
        // launch before it was issued into a busy one -- is still issued as one of a stream)
        const bool busy = piped && other_launch_in_flight(s, &L);
        // (the hint is about CALLS: a call's later batches always find its earlier ones in flight, which says nothing about
        // whether the caller streams frames or waits for each. The batches of a call that found the GPU idle are sized by
        // the batches behind them, the last one takes the full grid and the isolated launch's list: it ends with nothing
        // behind it -- a blocking 1920x1080x512 eighth, two batches: 14.7 ms; both on the full grid 14.8; the FIRST on the
        // full grid and the second on 12 of 16 slots 15.8; the second on the stream's 3: 38.6)
        if (b == 0) call_streams = busy || (piped && s->streaming_hint);
        const uint32_t to_come = uint32_t(n_batches - 1 - b);
        const bool overlapped = call_streams || (piped && to_come != 0u);
        // (a blocking call's batch shares the GPU with the batches behind it: the last one ends alone and takes the full
        // grid, the one before it shares with one, ...; a stream's launch takes the steady state's share: grid_for)
        const uint32_t company = to_come;
        if (b == 0) s->streaming_hint = busy;
        P.work_stripes = !overlapped ? s->work_stripes
                         : s->work_stripes_overlap != kStripesAuto ? s->work_stripes_overlap
                         : 0u;  // (contiguous shards; with half grids three deep small launches preferred stripes of 4, round 3)
        P.sample_buf = B.d_sample_buf;
        P.work_counter = B.d_work_counter;
        P.gseq = L.d_gseq;
        P.gstack = L.d_gstack;
        P.helper_words = L.d_helper_words, P.helper_seq = ++L.seq, P.wave_base = 0u;
        // which of the lane's two sets of tile tables this launch reads: the one made for this camera and partition, else
        // the one used longer ago, filled now
        rbrt_hip_scene::Lane::TileSet* S = nullptr;
        if (tile_pass) {
            rbrt_hip_scene::Lane::TileKey key;
            std::memset(&key, 0, sizeof(key));
            const uint32_t list_mode = list_mode_for(s, overlapped && !stats);
            key.cam = *cam, key.rank = o->tile_rank, key.world = world, key.list_mode = list_mode, key.min_dist = o->min_dist;
            for (auto& C : L.tiles)
                if (C.key_valid && std::memcmp(&key, &C.key, sizeof(key)) == 0) S = &C;
            if (!S) {
                S = L.tiles[0].last_used <= L.tiles[1].last_used ? &L.tiles[0] : &L.tiles[1];
                // behind the set's last readers and behind the pass that wrote it last (whatever stream that ran on), on the
                // prep stream: beside the launches in flight, ahead of this one
                // (a launch that has the GPU to itself has nothing to run beside: its pass goes on its own stream, one
                // cross-stream hop less in front of a blocking frame)
                if (overlapped)
                    if (int rc = ensure_prep_stream()) return rc;
                hipStream_t cs = overlapped ? s->prep_stream : ts;
                if (S->free_recorded) HIP_TRY(hipStreamWaitEvent(cs, S->ev_free, 0));
                if (S->key_valid) HIP_TRY(hipStreamWaitEvent(cs, S->ev_lists, 0));
                P.tile_cull = S->d_cull, P.tile_lists = S->d_lists;
                P.tile_list_mode = list_mode;
                P.tiles_reversed = rowmajor_reversed;  // (mode 4 builds its list in the direction row-major order is handed out in)
                P.tile_lists_wide = overlapped ? 0u : 1u;
                HIP_TRY(launch_primary_cull(P, cs));
                HIP_TRY(hipEventRecord(S->ev_lists, cs));
                S->key = key, S->key_valid = true;
                if (s->trace_launches) std::fprintf(stderr, "[rbrt_hip] tile pass on the prep stream: lane %u set %d list_mode %u\n",
                                                    unsigned(&L - s->lanes.data()), int(S - L.tiles), list_mode);
            }
            S->last_used = ++s->launch_no;
            if (s->trace_launches)
                std::fprintf(stderr, "[rbrt_hip] t %.3f ms launch %llu lane %u %s grid list_mode %u set %d\n", now_s() * 1e3, (unsigned long long)s->launch_no,
                             unsigned(&L - s->lanes.data()), overlapped ? "half" : "full", S->key.list_mode, int(S - L.tiles));
        }
        P.tile_cull = S ? S->d_cull : nullptr;
        P.tile_lists = S ? S->d_lists : nullptr;
        P.tile_list_mode = S ? S->key.list_mode : 0u;
        if (S && S->key.list_mode != 0u) P.tiles_reversed = 0u;  // (the list is in hand-out order already)
        else if (S) P.tiles_reversed = rowmajor_reversed;
        // (B.d_work_counter is zero: from its allocation, afterwards from the resolve kernel of the launch that used the set last)
        // RBRT_POISON_SAMPLES=1 (tests): a (pixel, sample) the kernel fails to write shows up as NaN in the image
        if (s->poison_samples) HIP_TRY(hipMemsetAsync(B.d_sample_buf, 0xFF, B.sample_buf_bytes, ts));
        if (timing) HIP_TRY(hipEventRecord(s->events[ev0 + 3 * b], ts));  // after the memset nodes
        if (S) HIP_TRY(hipStreamWaitEvent(ts, S->ev_lists, 0));  // (the set's tables: made on the prep stream, or at the first call on the caller's)
        const uint32_t grid = stats ? s->n_waves : grid_for(s, overlapped, depth, call_streams, company, P.n_items);
        (grid < s->n_waves ? s->n_half_grid : s->n_full_grid) += 1;
        const bool share_build = s->share_idle != 0u && P.n_items < s->share_below;
        if (piped) HIP_TRY(hipEventRecord(L.ev_ready, ts));  // (behind everything the launch waits for: a helper launch waits for this)
        HIP_TRY(launch_trace_megakernel(P, grid, s->pool, stats, share_build, ts));
        if (piped && s->pool == 128u && s->helpers_mode != 0u) {
            L.open.valid = true, L.open.P = P, L.open.grid = grid, L.open.helper_waves = 0u, L.open.rounds = 0u, L.open.share = share_build;
            L.open.seq = ++s->open_seq;
            if (s->helpers_mode == 2u) {  // (tests: a helper with every overlapped launch, on a stream of its own)
                if (!s->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking));
                if (int rc = issue_helper(s, L, s->n_cus, s->aux_stream)) return rc;
            }
        }
        if (timing) HIP_TRY(hipEventRecord(s->events[ev0 + 3 * b + 1], ts));
        R.sample_buf = B.d_sample_buf;
        R.work_counter = B.d_work_counter;
        R.batch = nb;
        R.first_batch = base == 0;
        R.last_batch = base + nb == o->spp;
        R.tile_lists = P.tile_lists;
        R.counters = stats ? s->d_counters : nullptr;
        R.helper_words = P.helper_words, R.helper_seq = P.helper_seq;
        R.error_flag = &s->d_counters->diag[57];
        if (piped) {
            HIP_TRY(hipEventRecord(L.ev_traced, ts));
            L.in_use = true;
            HIP_TRY(hipStreamWaitEvent(stream, L.ev_traced, 0));
        }
        HIP_TRY(launch_resolve(R, stream));
        // The background-only tiles never reach the trace kernel: their pixels are finished here, on the caller's stream
        // like every write to its buffers. BEHIND the resolve although it needs only the lists: issued beside its own
        // trace launch it waited for wave slots that persistent trace waves hold until their launch ends (1.2 ms on
        // average for 0.05 ms of work, with the resolve queued behind it); now it runs in the slots that launch just freed.
        if (tile_pass) HIP_TRY(launch_sky_resolve(P, R, stream));
        if (S) {  // the set's last readers have been issued: a later pass into it waits for them
            HIP_TRY(hipEventRecord(S->ev_free, stream));
            S->free_recorded = true;
        }
        if (depth > 1) {  // (also after a counting launch: it used lane 0's buffers on the caller's stream)
            HIP_TRY(hipEventRecord(B.ev_resolved, stream));
            B.in_use = true;
        }
        if (timing) {
            HIP_TRY(hipEventRecord(s->events[ev0 + 3 * b + 2], stream));
            s->events_used = ev0 + 3 * (b + 1);
        }
    }
    // the watcher of elastic launches: started by the first call of a stream, woken when there is a launch to look after
    s->last_call_s = now_s();
    if (s->helpers_mode == 1u && depth > 1 && !stats) {
        if (!s->watcher.joinable() && streams_now) {
            try {
                s->watcher = std::thread([s]() {
                    try {
                        watcher_main(s);
                    } catch (...) {  // (nothing the watcher does is needed: launches then stay the size they were issued with)
                    }
                });
            } catch (...) {  // (no thread to be had: launches stay the size they were issued with)
            }
        }
        s->cv.notify_one();
    }
    return RBRT_OK;
}

int rbrt_hip_render_device(rbrt_hip_scene_t* s, const rbrt_camera_t* cam, const rbrt_render_opts_t* o,
                           void* stream, float* d_radiance, uint8_t* d_rgb8) {
    return render_samples(s, cam, o, stream, 0, o ? o->spp : 0, nullptr, d_radiance, d_rgb8);
}

int rbrt_hip_render_pass(rbrt_hip_scene_t* s, const rbrt_camera_t* cam, const rbrt_render_opts_t* o, void* stream,
                         uint32_t sample_begin, uint32_t sample_end, float* d_accum, float* d_radiance, uint8_t* d_rgb8) {
    if (!d_accum) return fail(RBRT_ERR_INVALID_ARG, "render_pass: d_accum is null");
    return render_samples(s, cam, o, stream, sample_begin, sample_end, d_accum, d_radiance, d_rgb8);
}

int rbrt_hip_scene_set_pipeline(rbrt_hip_scene_t* s, uint32_t depth) {
    if (!s) return fail(RBRT_ERR_INVALID_ARG, "set_pipeline: null scene");
    if (depth > kMaxPipeline) return fail(RBRT_ERR_INVALID_ARG, "set_pipeline: depth must be 0 (automatic) or 1..8");
    HIP_TRY(hipSetDevice(s->device));
    std::lock_guard<std::mutex> watcher_lock(s->mu);
    HIP_TRY(hipDeviceSynchronize());
    s->pipeline = depth;
    s->next_lane = 0;
    return ensure_lanes(s, depth ? depth : 2u);
}

int rbrt_hip_scene_stats(rbrt_hip_scene_t* s, rbrt_hip_stats_t* out) {
    if (!s || !out) return fail(RBRT_ERR_INVALID_ARG, "stats: null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    DevCounters c;
    HIP_TRY(hipMemcpy(&c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    s->stats.rays = c.rays;
    s->stats.mesh_gate_pass = c.mesh_gate_pass;
    s->stats.nodes_visited = c.nodes_visited;
    s->stats.tris_tested = c.tris_tested;
    s->stats.mesh_hits = c.mesh_hits;
    s->stats.samples = c.samples;
    s->stats.nan_discriminants = c.nan_discriminants;
    s->stats.node_bytes = sizeof(BvhNode4);
    s->stats.tri_bytes = sizeof(BvhTri);
    *out = s->stats;
    if (c.diag[57])
        return fail(RBRT_ERR_HIP, "the trace kernel detected corrupt internal state (a path slot without a valid sample index)");
    return RBRT_OK;
}

// Synchronises, then reports and clears what the kernels flagged since the last check: a NaN sphere discriminant
// (the reference panics there, sphere.rs:33) or a path slot without a valid sample index (internal corruption).
int rbrt_hip_scene_check(rbrt_hip_scene_t* s) {
    if (!s) return fail(RBRT_ERR_INVALID_ARG, "check: null scene");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long nan = 0, corrupt = 0;
    HIP_TRY(hipMemcpy(&nan, &s->d_counters->nan_discriminants, sizeof(nan), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&corrupt, &s->d_counters->diag[57], sizeof(corrupt), hipMemcpyDeviceToHost));
    if (nan) HIP_TRY(hipMemset(&s->d_counters->nan_discriminants, 0, sizeof(nan)));
    if (corrupt) HIP_TRY(hipMemset(&s->d_counters->diag[57], 0, sizeof(corrupt)));
    if (corrupt)
        return fail(RBRT_ERR_HIP, "the trace kernel detected corrupt internal state (a path slot without a valid sample index)");
    if (nan)
        return fail(RBRT_ERR_NAN, "a sphere discriminant was NaN " + std::to_string(nan) +
                                      " times (the reference panics: sphere.rs:33); those rays were treated as misses");
    return RBRT_OK;
}

int rbrt_hip_scene_create_times(rbrt_hip_scene_t* s, rbrt_hip_call_times_t* out) {
    if (!s || !out) return fail(RBRT_ERR_INVALID_ARG, "create_times: null argument");
    *out = s->create_times;
    return RBRT_OK;
}

int rbrt_hip_last_render_times(rbrt_hip_call_times_t* out) {
    if (!out) return fail(RBRT_ERR_INVALID_ARG, "last_render_times: null argument");
    *out = g_last_render_times;
    return RBRT_OK;
}

// Waits (at most timeout_s) for the background build of a scene's trees (struct Refine) and adopts its result.
// *state: 0 = no background build was started (host-built meshes, a forced builder, RBRT_BVH_REFINE=0, a one-shot call),
// 1 = its trees are in use now, 2 = it failed or was cancelled (the device builder's trees stay), 3 = still at work.
int rbrt_hip_scene_refine_wait(rbrt_hip_scene_t* s, double timeout_s, int* state, double* build_seconds) {
    if (!s) return fail(RBRT_ERR_INVALID_ARG, "refine_wait: null scene");
    int st = 0;
    if (Refine* r = s->refine.get()) {
        const double t0 = now_s();
        while (r->state.load(std::memory_order_acquire) == 0 && now_s() - t0 < timeout_s) std::this_thread::sleep_for(std::chrono::microseconds(200));
        HIP_TRY(hipSetDevice(s->device));
        adopt_refined(s);
        const int rs = r->state.load(std::memory_order_acquire);
        st = rs == 0 ? 3 : s->n_refined != 0 ? 1 : 2;
        if (build_seconds) *build_seconds = rs == 0 ? 0.0 : r->seconds;
        if (st == 2 && r->th.joinable()) r->th.join();
        if (st == 2) g_last_error = "background BVH build: " + r->error;
    } else if (build_seconds) {
        *build_seconds = 0.0;
    }
    if (state) *state = st;
    return RBRT_OK;
}

int rbrt_hip_scene_info(rbrt_hip_scene_t* s, rbrt_hip_scene_info_t* out) {
    if (!s || !out) return fail(RBRT_ERR_INVALID_ARG, "scene_info: null argument");
    adopt_refined(s);
    std::memset(out, 0, sizeof(*out));
    out->n_spheres = s->n_spheres, out->n_meshes = s->n_meshes;
    out->n_meshes_device_built = s->n_device_built;
    out->bvh_stack_need = s->stack_need;
    out->n_nodes = s->total_nodes, out->n_triangles = s->total_tris;
    out->trace_waves = s->n_waves;
    out->lds_bytes_per_wave = uint32_t(megakernel_lds_bytes(s->pool, s->stack_entries, s->n_spheres, s->n_meshes, s->n_elem_tris));
    (void)hipSetDevice(s->device);
    out->occupancy_api_waves_per_cu = uint32_t(megakernel_occupancy_per_cu(s->pool, out->lds_bytes_per_wave));
    out->n_cus = s->n_cus;
    return RBRT_OK;
}

int rbrt_hip_scene_last_batching(rbrt_hip_scene_t* s, uint32_t* samples_per_batch, uint32_t* n_batches) {
    if (!s) return fail(RBRT_ERR_INVALID_ARG, "last_batching: null scene");
    if (samples_per_batch) *samples_per_batch = s->last_batch;
    if (n_batches) *n_batches = s->last_n_batches;
    return RBRT_OK;
}

int rbrt_hip_unpack_tiles_strided(int device, void* stream, const float* d_gathered, uint32_t width, uint32_t height,
                                  uint32_t tile_world, size_t rank_stride_pixels, float* d_radiance, uint8_t* d_rgb8) {
    if (!d_gathered) return fail(RBRT_ERR_INVALID_ARG, "unpack: null input");
    const uint32_t world = tile_world ? tile_world : 1;
    if (rank_stride_pixels != 0 && rank_stride_pixels < rbrt_hip_packed_pixels(width, height, 0, world))
        return fail(RBRT_ERR_INVALID_ARG, "unpack: rank stride smaller than rank 0's packed tiles");
    if (int rc = ensure_device(device)) return rc;
    HIP_TRY(launch_unpack(d_gathered, width, height, world, rank_stride_pixels, d_radiance, d_rgb8,
                          static_cast<hipStream_t>(stream)));
    return RBRT_OK;
}

int rbrt_hip_unpack_tiles(int device, void* stream, const float* d_gathered, uint32_t width, uint32_t height,
                          uint32_t tile_world, float* d_radiance, uint8_t* d_rgb8) {
    return rbrt_hip_unpack_tiles_strided(device, stream, d_gathered, width, height, tile_world, 0, d_radiance, d_rgb8);
}

int rbrt_hip_render(const rbrt_camera_t* cam, const rbrt_scene_t* scene, const rbrt_render_opts_t* opts,
                    float* out_radiance, uint8_t* out_rgb8) {
    if (!cam || !scene || !opts) return fail(RBRT_ERR_INVALID_ARG, "render: null argument");
    const double t_call0 = now_s();
    rbrt_hip_call_times_t& times = g_last_render_times;
    std::memset(&times, 0, sizeof(times));
    rbrt_hip_scene_t* s = nullptr;
    // One render, then the scene is gone again: its trees are made by whichever builder costs THIS call less (a frame of
    // ~10 G path samples per second against the host builder's ~1 us per triangle: device_builder_is_cheaper), no
    // background build is started, and it gets the lanes its sample batches will use, not a stream's eight.
    CreateHint hint;
    hint.one_shot = true;
    {
        const uint32_t w1 = opts->tile_world ? opts->tile_world : 1;
        hint.render_s_est = double(cam->img_width_pix) * cam->img_height_pix * opts->spp / w1 / 10.0e9;
    }
    if (int rc = scene_create_impl(scene, 0, &s, hint)) return rc;
    times = s->create_times;
    if (s->pipeline == 0) s->pipeline = 3;  // (the batches of this one render overlap on three lanes)
    const double t_render0 = now_s();
    const uint32_t world = opts->tile_world ? opts->tile_world : 1;
    const size_t npix_img = size_t(cam->img_width_pix) * cam->img_height_pix;
    const size_t n_out = world > 1
                             ? rbrt_hip_packed_pixels(cam->img_width_pix, cam->img_height_pix, opts->tile_rank, world)
                             : npix_img;
    float* d_rad = nullptr;
    uint8_t* d_rgb = nullptr;
    int rc = RBRT_OK;
    auto cleanup = [&]() { rbrt_hip_scene_destroy(s); };  // (the images are the scene's memory: dev_alloc)
#define TRY_OR_CLEAN(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            cleanup();                                                                      \
            return fail(_e == hipErrorOutOfMemory ? RBRT_ERR_OOM : RBRT_ERR_HIP,            \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                 \
        }                                                                                   \
    } while (0)
    if (n_out == 0) {
        cleanup();
        return RBRT_OK;
    }
    {
        void* p = nullptr;
        const size_t rad_bytes = (n_out * 3 * sizeof(float) + kSlabAlign - 1) & ~(kSlabAlign - 1);
        TRY_OR_CLEAN(dev_alloc(s, rad_bytes + (out_rgb8 ? n_out * 3 : 0), &p));
        d_rad = static_cast<float*>(p);
        if (out_rgb8) d_rgb = static_cast<uint8_t*>(p) + rad_bytes;
    }
    rc = rbrt_hip_render_device(s, cam, opts, nullptr, d_rad, d_rgb);
    if (rc) {
        cleanup();
        return rc;
    }
    TRY_OR_CLEAN(hipDeviceSynchronize());
    const double t_copy0 = now_s();
    times.render_s = t_copy0 - t_render0;
    if (world > 1) {
        // scatter this rank's packed tiles into the caller's full-size images
        std::vector<float> hr(n_out * 3);
        std::vector<uint8_t> hb(out_rgb8 ? n_out * 3 : 0);
        TRY_OR_CLEAN(hipMemcpy(hr.data(), d_rad, hr.size() * sizeof(float), hipMemcpyDeviceToHost));
        if (out_rgb8) TRY_OR_CLEAN(hipMemcpy(hb.data(), d_rgb, hb.size(), hipMemcpyDeviceToHost));
        const uint32_t W = cam->img_width_pix, H = cam->img_height_pix;
        const uint32_t tiles_x = (W + RBRT_TILE - 1) / RBRT_TILE;
        const size_t n_local = n_out / 64;
        for (size_t tl = 0; tl < n_local; ++tl) {
            const uint32_t tile = uint32_t(tl) * world + opts->tile_rank;
            uint32_t ty, tx;
            tile_xy(tile, tiles_x, ty, tx);
            for (uint32_t p = 0; p < 64; ++p) {
                const uint32_t row = ty * RBRT_TILE + p / 8, col = tx * RBRT_TILE + p % 8;
                if (row >= H || col >= W) continue;
                const size_t src = (tl * 64 + p) * 3, dst = (size_t(row) * W + col) * 3;
                for (int c = 0; c < 3; ++c) {
                    if (out_radiance) out_radiance[dst + c] = hr[src + c];
                    if (out_rgb8) out_rgb8[dst + c] = hb[src + c];
                }
            }
        }
    } else {
        if (out_radiance)
            TRY_OR_CLEAN(hipMemcpy(out_radiance, d_rad, n_out * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (out_rgb8) TRY_OR_CLEAN(hipMemcpy(out_rgb8, d_rgb, n_out * 3, hipMemcpyDeviceToHost));
    }
    DevCounters c;
    TRY_OR_CLEAN(hipMemcpy(&c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    const double t_destroy0 = now_s();
    times.copy_s = t_destroy0 - t_copy0;
    cleanup();
    times.destroy_s = now_s() - t_destroy0;
    times.total_s = now_s() - t_call0;
#undef TRY_OR_CLEAN
    if (c.diag[57])
        return fail(RBRT_ERR_HIP, "the trace kernel detected corrupt internal state (a path slot without a valid sample index)");
    if (c.nan_discriminants)
        return fail(RBRT_ERR_NAN, "a sphere discriminant was NaN (the reference panics: sphere.rs:33); "
                                  "those rays were treated as misses");
    return RBRT_OK;
}

// Diagnostic: the megakernel's pass statistics of the last counting render (layout: DevCounters::diag).
int rbrt_hip_scene_debug_counters(rbrt_hip_scene_t* s, uint64_t* out, size_t n) {
    if (!s || !out) return fail(RBRT_ERR_INVALID_ARG, "debug_counters: null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    DevCounters c;
    HIP_TRY(hipMemcpy(&c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n && i < 64; ++i) out[i] = c.diag[i];
    return RBRT_OK;
}

// Diagnostic: preset one diag slot (the analysis builds keep minima there, which have to start at all-ones).
int rbrt_hip_scene_debug_set_counter(rbrt_hip_scene_t* s, size_t index, uint64_t value) {
    if (!s || index >= 64) return fail(RBRT_ERR_INVALID_ARG, "debug_set_counter: bad argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    const unsigned long long v = value;
    HIP_TRY(hipMemcpy(&s->d_counters->diag[index], &v, sizeof(v), hipMemcpyHostToDevice));
    return RBRT_OK;
}

// Diagnostic: run the host BVH builder alone (no device needed) and hand back copies of its arrays.
int rbrt_hip_bvh_build_host(const rbrt_mesh_t* mesh, void** nodes_out, size_t* n_nodes, void** tris_out,
                            size_t* n_tris, uint32_t* max_depth, float* max_e12) {
    if (!mesh || !nodes_out || !n_nodes || !tris_out || !n_tris)
        return fail(RBRT_ERR_INVALID_ARG, "bvh_build_host: null argument");
    BvhBuildResult r = build_bvh(*mesh);
    *n_nodes = r.nodes.size();
    *n_tris = r.tris.size();
    *nodes_out = std::malloc(std::max<size_t>(1, r.nodes.size() * sizeof(BvhNode4)));
    *tris_out = std::malloc(std::max<size_t>(1, r.tris.size() * sizeof(BvhTri)));
    if (!*nodes_out || !*tris_out) return fail(RBRT_ERR_OOM, "bvh_build_host: malloc failed");
    std::memcpy(*nodes_out, r.nodes.data(), r.nodes.size() * sizeof(BvhNode4));
    std::memcpy(*tris_out, r.tris.data(), r.tris.size() * sizeof(BvhTri));
    if (max_depth) *max_depth = r.max_depth;
    if (max_e12) *max_e12 = r.max_e12;
    return RBRT_OK;
}
void rbrt_hip_free_host(void* p) { std::free(p); }

// Diagnostic: the host builder over n 48-B triangle records in any order (what the background thread of a scene handle
// runs on the device builder's output, struct Refine). Same outputs as rbrt_hip_bvh_build_host.
int rbrt_hip_bvh_build_host_records(const void* records, size_t n, void** nodes_out, size_t* n_nodes, void** tris_out,
                                    size_t* n_tris, uint32_t* max_depth, float* max_e12) {
    if ((!records && n) || !nodes_out || !n_nodes || !tris_out || !n_tris)
        return fail(RBRT_ERR_INVALID_ARG, "bvh_build_host_records: null argument");
    BvhBuildResult r = build_bvh_from_records(static_cast<const BvhTri*>(records), n);
    *n_nodes = r.nodes.size();
    *n_tris = r.tris.size();
    *nodes_out = std::malloc(std::max<size_t>(1, r.nodes.size() * sizeof(BvhNode4)));
    *tris_out = std::malloc(std::max<size_t>(1, r.tris.size() * sizeof(BvhTri)));
    if (!*nodes_out || !*tris_out) return fail(RBRT_ERR_OOM, "bvh_build_host_records: malloc failed");
    std::memcpy(*nodes_out, r.nodes.data(), r.nodes.size() * sizeof(BvhNode4));
    std::memcpy(*tris_out, r.tris.data(), r.tris.size() * sizeof(BvhTri));
    if (max_depth) *max_depth = r.max_depth;
    if (max_e12) *max_e12 = r.max_e12;
    return RBRT_OK;
}

// Diagnostic: the GPU builder alone (bvh_device.hip), results copied back in the layout of rbrt_hip_bvh_build_host.
// *built = 0 when the builder declined the mesh (tiny, or a tree beyond the depth budget); nothing is returned then.
int rbrt_hip_bvh_build_device(const rbrt_mesh_t* mesh, void** nodes_out, size_t* n_nodes, void** tris_out, size_t* n_tris,
                              uint32_t* max_depth, float* max_e12, int* built) {
    if (!mesh || !nodes_out || !n_nodes || !tris_out || !n_tris || !built)
        return fail(RBRT_ERR_INVALID_ARG, "bvh_build_device: null argument");
    *built = 0, *nodes_out = *tris_out = nullptr, *n_nodes = *n_tris = 0;
    if (int rc = ensure_device(0)) return rc;
    if (mesh->n_total < 8) return RBRT_OK;
    BvhTri* d_tris = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_tris), size_t(mesh->n_total) * sizeof(BvhTri)));
    DeviceBvhResult r;
    hipError_t e = device_build_mesh(*mesh, d_tris, 0, nullptr, &r);
    if (e == hipSuccess && r.ok) {
        *nodes_out = std::malloc(size_t(r.n_nodes) * sizeof(BvhNode4));
        *tris_out = std::malloc(std::max<size_t>(1, size_t(r.n_valid) * sizeof(BvhTri)));
        if (*nodes_out && *tris_out) {
            e = hipMemcpy(*nodes_out, r.d_nodes, size_t(r.n_nodes) * sizeof(BvhNode4), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(*tris_out, d_tris, size_t(r.n_valid) * sizeof(BvhTri), hipMemcpyDeviceToHost);
            *n_nodes = r.n_nodes, *n_tris = r.n_valid, *built = 1;
            if (max_depth) *max_depth = r.max_depth;
            if (max_e12) *max_e12 = r.max_e12;
        }
    }
    if (r.d_nodes) (void)hipFree(r.d_nodes);
    (void)hipFree(d_tris);
    if (e != hipSuccess) return fail(RBRT_ERR_HIP, std::string("bvh_build_device: ") + hipGetErrorString(e));
    return RBRT_OK;
}

// Test hook: BoundingBox::hit (aabbox.rs:28-58) on n rays against one box, by the division-free form the megakernel
// uses (out_fast) and by the verbatim IEEE form (out_exact). Host arrays.
int rbrt_hip_selftest_gate(const float lo[3], const float hi[3], const float* rays, size_t n, uint8_t* out_fast,
                           uint8_t* out_exact) {
    if (!lo || !hi || (!rays && n) || !out_fast || !out_exact) return fail(RBRT_ERR_INVALID_ARG, "selftest_gate: null argument");
    if (n == 0) return RBRT_OK;
    if (int rc = ensure_device(0)) return rc;
    float *d_box = nullptr, *d_rays = nullptr;
    uint8_t *d_f = nullptr, *d_e = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_box), (void)hipFree(d_rays), (void)hipFree(d_f), (void)hipFree(d_e); };
    const float box[6] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_box), sizeof(box));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_rays), n * 6 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_f), n);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_e), n);
    if (e == hipSuccess) e = hipMemcpy(d_box, box, sizeof(box), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, n * 6 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_gate_selftest(d_box, d_rays, n, d_f, d_e);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out_fast, d_f, n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_exact, d_e, n, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(RBRT_ERR_HIP, std::string("selftest_gate: ") + hipGetErrorString(e));
    return RBRT_OK;
}

// Test hook: the kernels' short forms of IEEE sqrt / normalize against the compiler's, on n pseudo-random operands.
int rbrt_hip_selftest_ieee(uint64_t seed, size_t n, uint64_t counts[3]) {
    if (!counts) return fail(RBRT_ERR_INVALID_ARG, "selftest_ieee: null argument");
    if (n >= (1ull << 32) * 256ull) return fail(RBRT_ERR_INVALID_ARG, "selftest_ieee: n too large");
    if (int rc = ensure_device(0)) return rc;
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, 3 * sizeof(unsigned long long));
    if (e == hipSuccess) e = launch_ieee_selftest(seed, n, d);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RBRT_ERR_HIP, std::string("selftest_ieee: ") + hipGetErrorString(e));
    for (int k = 0; k < 3; ++k) counts[k] = h[k];
    return RBRT_OK;
}

// Test hook: n scatter events (materials.rs:4-12) on the device functions the shading passes use. Host arrays.
int rbrt_hip_debug_scatter(const rbrt_material_t* mats, const float* in_dir, const float* p, const float* normal,
                           const uint32_t* rng_state, size_t n, float* out_dir, uint8_t* out_ok, uint32_t* out_rng_state) {
    if (n == 0) return RBRT_OK;
    if (!mats || !in_dir || !p || !normal || !rng_state) return fail(RBRT_ERR_INVALID_ARG, "debug_scatter: null argument");
    std::vector<DevMaterial> hm(n);
    for (size_t i = 0; i < n; ++i) {
        if (int rc = check_material(mats[i])) return rc;
        for (int c = 0; c < 3; ++c) hm[i].albedo[c] = mats[i].albedo[c];
        hm[i].param = mats[i].param, hm[i].kind = mats[i].kind;
    }
    if (int rc = ensure_device(0)) return rc;
    DevMaterial* d_m = nullptr;
    float *d_in = nullptr, *d_p = nullptr, *d_n = nullptr, *d_out = nullptr;
    uint32_t *d_rng = nullptr, *d_rng_out = nullptr;
    uint8_t* d_ok = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_m), (void)hipFree(d_in), (void)hipFree(d_p), (void)hipFree(d_n), (void)hipFree(d_out), (void)hipFree(d_rng),
            (void)hipFree(d_rng_out), (void)hipFree(d_ok);
    };
    const size_t v3 = n * 3 * sizeof(float), st = n * 2 * sizeof(uint32_t);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_m), n * sizeof(DevMaterial));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_in), v3);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_p), v3);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_n), v3);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_out), v3);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_rng), st);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_rng_out), st);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_ok), n);
    if (e == hipSuccess) e = hipMemcpy(d_m, hm.data(), n * sizeof(DevMaterial), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_in, in_dir, v3, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_p, p, v3, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_n, normal, v3, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_rng, rng_state, st, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_scatter_debug(d_m, d_in, d_p, d_n, d_rng, n, d_out, d_ok, d_rng_out);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && out_dir) e = hipMemcpy(out_dir, d_out, v3, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_ok) e = hipMemcpy(out_ok, d_ok, n, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_rng_state) e = hipMemcpy(out_rng_state, d_rng_out, st, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(RBRT_ERR_HIP, std::string("debug_scatter: ") + hipGetErrorString(e));
    return RBRT_OK;
}

int rbrt_hip_debug_primary_cull(rbrt_hip_scene_t* s, const rbrt_camera_t* cam, uint32_t* out_words, size_t n_words) {
    if (!s || !cam || !out_words) return fail(RBRT_ERR_INVALID_ARG, "debug_primary_cull: null argument");
    const uint32_t tiles_x = (cam->img_width_pix + RBRT_TILE - 1) / RBRT_TILE, tiles_y = (cam->img_height_pix + RBRT_TILE - 1) / RBRT_TILE;
    if (uint64_t(tiles_x) * tiles_y != n_words || n_words == 0 || n_words > 0xFFFFFFFFull)
        return fail(RBRT_ERR_INVALID_ARG, "debug_primary_cull: n_words must be the number of 8x8 tiles of the camera's image");
    HIP_TRY(hipSetDevice(s->device));
    rbrt_render_opts_t o;
    rbrt_render_opts_default(&o);
    TraceParams P;
    fill_trace_params(s, cam, &o, P);
    P.tiles_x = tiles_x, P.tiles_y = tiles_y, P.n_tiles = uint32_t(n_words);
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, n_words * sizeof(uint32_t)));
    P.tile_cull = static_cast<uint32_t*>(d);
    hipError_t e = launch_primary_cull(P, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    std::vector<uint32_t> by_number(n_words);
    if (e == hipSuccess) e = hipMemcpy(by_number.data(), d, n_words * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RBRT_ERR_HIP, std::string("debug_primary_cull: ") + hipGetErrorString(e));
    // (the table is indexed by tile NUMBER, rbrt_hip.h "How tiles are dealt to ranks"; the hook hands it over in image order)
    for (uint32_t ty = 0; ty < tiles_y; ++ty)
        for (uint32_t tx = 0; tx < tiles_x; ++tx) out_words[size_t(ty) * tiles_x + tx] = by_number[tile_number(ty, tx, tiles_x)];
    return RBRT_OK;
}

int rbrt_hip_trace_rays(rbrt_hip_scene_t* s, const float* rays, size_t n, float min_dist, float max_dist,
                        float* out_t, int32_t* out_obj, int32_t* out_tri, float* out_dist) {
    if (!s || (!rays && n)) return fail(RBRT_ERR_INVALID_ARG, "trace_rays: null argument");
    if (n == 0) return RBRT_OK;
    HIP_TRY(hipSetDevice(s->device));
    rbrt_render_opts_t o;
    rbrt_render_opts_default(&o);
    o.min_dist = min_dist;
    o.max_dist = max_dist;
    TraceParams P;
    fill_trace_params(s, nullptr, &o, P);
    float *d_rays = nullptr, *d_t = nullptr, *d_dist = nullptr;
    int32_t *d_obj = nullptr, *d_tri = nullptr;
    int rc = RBRT_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_rays), (void)hipFree(d_t), (void)hipFree(d_dist), (void)hipFree(d_obj), (void)hipFree(d_tri);
    };
#define TRY_OR_CLEAN(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            cleanup();                                                                           \
            return fail(RBRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
        }                                                                                        \
    } while (0)
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_rays), n * 6 * sizeof(float)));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_t), n * sizeof(float)));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_dist), n * sizeof(float)));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_obj), n * sizeof(int32_t)));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_tri), n * sizeof(int32_t)));
    TRY_OR_CLEAN(hipMemcpy(d_rays, rays, n * 6 * sizeof(float), hipMemcpyHostToDevice));
    TRY_OR_CLEAN(launch_trace_rays(P, d_rays, n, d_t, d_obj, d_tri, d_dist, nullptr));
    TRY_OR_CLEAN(hipDeviceSynchronize());
    if (out_t) TRY_OR_CLEAN(hipMemcpy(out_t, d_t, n * sizeof(float), hipMemcpyDeviceToHost));
    if (out_dist) TRY_OR_CLEAN(hipMemcpy(out_dist, d_dist, n * sizeof(float), hipMemcpyDeviceToHost));
    if (out_obj) TRY_OR_CLEAN(hipMemcpy(out_obj, d_obj, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out_tri) TRY_OR_CLEAN(hipMemcpy(out_tri, d_tri, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    cleanup();
#undef TRY_OR_CLEAN
    return rc;
}

}  // extern "C"
