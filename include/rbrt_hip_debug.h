/*
 * rbrt_hip_debug.h — test hooks, diagnostics and lab knobs of librbrt_hip.so.
 *
 * NOT part of the drop-in boundary (that is rbrt_hip.h, which a Rust host binds): nothing here is needed to
 * replace rbrt_lib::render_scene (rbrt_lib/src/lib.rs:75-79). These entry points exist so that the parity tests
 * can look inside the hot path (one Scene::hit, one scatter event, the BVH builders alone, the two forms of the
 * mesh gate) and so that bench.py can time the trace kernel with events on the stream it really runs on.
 *
 * Lab knobs (environment, read by rbrt_hip_scene_create ONLY when RBRT_HIP_LAB=1; a value outside the stated range
 * makes scene_create fail with RBRT_ERR_INVALID_ARG instead of being clamped; none of them changes the image):
 *   RBRT_POOL=128|256            path slots per wave            RBRT_LDS_STACK=1..64       stack entries per lane in LDS
 *   RBRT_Y_LOW / RBRT_Y_HIGH=1..64, RBRT_Y_HIGH_PARKED=1..256   refill water marks of the traversal lanes
 *   RBRT_LEAF_ROUND=1..64, RBRT_LEAF_LEAVES=1..128              when a leaf round runs
 *   RBRT_SHARE_IDLE=0..64, RBRT_SHARE_BELOW=<samples>           shared traversals in the drain
 *   RBRT_DRAIN_MODE=<bits 0,1,3>  RBRT_WORK_STRIPES, RBRT_WORK_STRIPES_OVERLAP=<chunks, power of two>
 *   RBRT_SHADE_ROUNDS=1..64, RBRT_SHADE_CONT_MIN=1..64          register-resident shading rounds
 *   RBRT_WAVES_PER_CU=1..32      RBRT_PIPELINE=0..8             RBRT_LANE_PRIORITY=low|default|high
 *   RBRT_BVH_CT=<SAH traversal cost, 4.0>   RBRT_PLOC_RADIUS=1..256 (device builder's neighbour search)
 *   RBRT_BVH_DEVICE_MIN=<entries>, RBRT_BVH_DEVICE_ALGO=ploc|lbvh   RBRT_POISON_SAMPLES=1 (tests)
 *   RBRT_PRIMARY_CULL=0|1        the tile pass (1): tiles whose camera rays reach nothing bypass the trace kernel
 *   RBRT_TILE_ORDER=0|1|2        tiles handed out: as api.cpp decides (0), first-to-last (1), last-to-first (2)
 *   RBRT_TILE_CLASSES=0..4       overrides the rule by launch kind: work list ascending (0), or by tile class: heavy|light,
 *                                light/2|heavy|light/2, light|heavy, or (4) row-major order with its last light tiles moved to the end
 *   RBRT_TILE_ISOLATED_MODE=0..4 the list mode of a launch that has the GPU to itself (4); a launch of a stream uses 0
 *   RBRT_TILE_TAIL_DIV=1..1024   mode 4: the share of the work list (1/n, 8) that is handed out last, from light tiles
 *   RBRT_OVERLAP_WAVES_PER_CU=0..16  waves per CU of a launch of a stream (0: 24 / launches side by side, rounded up, and
 *                                4 instead of 3 for a launch of 8 M work items or more)
 *   RBRT_TRACE_LAUNCHES=1        one stderr line per trace launch, tile pass and helper launch (which lane, grid, table set)
 *   RBRT_HELPERS=0|1|2           helper launches (elastic launches): never, by the watcher (1), one with every overlapped launch (tests)
 *   RBRT_HELPER_MIN_ITEMS=<n>    a helper wave joins only while n work items per wave are left (4096); RBRT_HELPER_ROUNDS=1..16 (4)
 *   RBRT_HELPER_MIN_LAUNCH_MI=<n> launches of n Mi work items or more get helper launches (16: smaller ones came out 1 % slower)
 *   RBRT_HELPER_MIN_FREE=1..16    helper launches only while this many wave slots per CU are free (1)
 *   RBRT_BVH_SPATIAL=0..0.6      the host builder's budget of duplicated references (spatial splits), as a share of the triangles
 *   RBRT_TRACE_CREATE=1          one stderr line per rbrt_hip_scene_create: where its time went
 */
#ifndef RBRT_HIP_DEBUG_H
#define RBRT_HIP_DEBUG_H

#include "rbrt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How the last rbrt_hip_render_device call on this scene split its samples: samples per batch (one trace launch
 * each, sized to the workspace cap $RBRT_HIP_WORKSPACE_MB and to the kernel's 32-bit work-item numbers) and the
 * number of batches. Diagnostic. */
int rbrt_hip_scene_last_batching(rbrt_hip_scene_t* scene, uint32_t* samples_per_batch, uint32_t* n_batches);

/* Test / diagnostic hook for Scene::hit (scene.rs:19-43): closest hit of n rays against the
 * resident scene. Host arrays. rays = n x {ox,oy,oz,dx,dy,dz}. Outputs (each may be NULL):
 *   out_t[n]      ray parameter of the winning object (NaN on miss)
 *   out_obj[n]    -1 on miss, sphere index in [0,n_spheres), or n_spheres + mesh index
 *   out_tri[n]    winning triangle index (reference numbering) for mesh hits, else -1
 *   out_dist[n]   dist_from_ray_orig of the winner (lib.rs:35)                            */
int rbrt_hip_trace_rays(rbrt_hip_scene_t* scene, const float* rays, size_t n, float min_dist,
                        float max_dist, float* out_t, int32_t* out_obj, int32_t* out_tri,
                        float* out_dist);

/* Test hook for BoundingBox::hit (aabbox.rs:28-58): n rays against the box [lo, hi], decided by the division-free
 * form the megakernel uses (out_fast[n]) and by the verbatim form with six IEEE divisions (out_exact[n]); the two
 * must agree for every input. Host arrays. */
int rbrt_hip_selftest_gate(const float lo[3], const float hi[3], const float* rays, size_t n, uint8_t* out_fast,
                           uint8_t* out_exact);

/* Test hook: the image needs correctly rounded sqrt and / (vec3.rs:111-126); the kernels use short forms of them when
 * every lane's operands are in the everyday range (kernels.hip "IEEE square root and division, the short way"). Runs n
 * pseudo-random operands through the short forms and the compiler's: counts[0] / counts[1] = differing sqrt results /
 * normalize components (must be 0), counts[2] = lanes that really took the short path. */
int rbrt_hip_selftest_ieee(uint64_t seed, size_t n, uint64_t counts[3]);

/* Diagnostic: pass statistics of the persistent megakernel from the last render with
 * RBRT_FLAG_COLLECT_STATS: out[0..5] passes per kind (empty, traverse, terminate, lambertian, metal,
 * dielectric), out[6..11] path slots handled per kind, out[12] traversal wave-steps, out[13] busy
 * lane-steps, out[14] refill rounds, out[15] scheduling rounds. */
int rbrt_hip_scene_debug_counters(rbrt_hip_scene_t* scene, uint64_t* out, size_t n);

/* Diagnostic: preset slot `index` (< 64) of the counters rbrt_hip_scene_debug_counters reads (analysis builds keep
 * minima there, which must start at all-ones). Synchronises the device. */
int rbrt_hip_scene_debug_set_counter(rbrt_hip_scene_t* scene, size_t index, uint64_t value);

/* Diagnostic: run the host-side BVH builder alone (needs no device). *nodes_out / *tris_out are
 * malloc'ed copies of the 128-B 4-wide node and 48-B triangle records (layout: rbrt_amd/csrc/device_types.h);
 * release them with rbrt_hip_free_host. */
int rbrt_hip_bvh_build_host(const rbrt_mesh_t* mesh, void** nodes_out, size_t* n_nodes, void** tris_out,
                            size_t* n_tris, uint32_t* max_depth, float* max_e12);
void rbrt_hip_free_host(void* p);
/* Diagnostic: the host builder over n 48-B triangle records (device_types.h BvhTri) in ANY order, `index` = the triangle's
 * position in the reference arrays: for the indexed records of a mesh, array for array the tree rbrt_hip_bvh_build_host
 * makes of that mesh. (It is what a scene handle's background thread runs on the device builder's output.) */
int rbrt_hip_bvh_build_host_records(const void* records, size_t n, void** nodes_out, size_t* n_nodes, void** tris_out,
                                    size_t* n_tris, uint32_t* max_depth, float* max_e12);
/* Diagnostic: the GPU-side BVH builder alone (rbrt_amd/csrc/bvh_device.hip; rbrt_hip_scene_create uses it wherever it costs the
 * call less than the host builder, $RBRT_BVH_BUILDER = host | device overrides). Same outputs as rbrt_hip_bvh_build_host;
 * *built = 0 when the builder declined the mesh (fewer than 8 entries, <= 4 indexed triangles, or a tree beyond the
 * traversal's depth budget), in which case scene_create falls back to the host builder. */
int rbrt_hip_bvh_build_device(const rbrt_mesh_t* mesh, void** nodes_out, size_t* n_nodes, void** tris_out,
                              size_t* n_tris, uint32_t* max_depth, float* max_e12, int* built);

/* Test hook for RayScattering::scatter (materials.rs:4-12; lambertian.rs:11-24, metal.rs:12-25, dielectric.rs:11-85):
 * n independent scatter events on the device code the megakernel shades with. Event i: incoming ray direction
 * in_dir[i], hit point p[i], hit normal normal[i] (as the intersection routines hand it over: unnormalised for
 * spheres), material mats[i], and the state of its random stream rng_state[i] = {s0, s1} (the xoroshiro64** words
 * of DESIGN.md "RNG contract"; the draws are consumed in the reference's order). Outputs (each may be NULL):
 *   out_dir[n][3]      direction of the scattered ray (its origin is p)
 *   out_ok[n]          the reference's bool (metal.rs:25 can return false)
 *   out_rng_state[n]   the stream's state afterwards (how many draws were consumed)
 * Host arrays. So that an image mismatch can be localised to one material without bisecting images. */
int rbrt_hip_debug_scatter(const rbrt_material_t* mats, const float* in_dir, const float* p, const float* normal,
                           const uint32_t* rng_state, size_t n, float* out_dir, uint8_t* out_ok, uint32_t* out_rng_state);

/* Test hook for the primary-ray culling table (DESIGN.md "The tile pass"): what the library computes on the trace
 * launch's stream before every launch for `cam`, one word per 8x8 tile of the image in row-major tile order
 * (n_words must be ceil(width/8) * ceil(height/8)): bit e < 24 = no camera ray of the tile can reach element e (spheres
 * only), bit 24 + m (m < 7) = none can pass the box of mesh m, bit 31 = the tile sees the background only (every sphere
 * out of reach, and every mesh: by its box, or by the boxes at the top of its tree). A set bit is a promise; tests check
 * it against the oracle's rays. Host array. */
int rbrt_hip_debug_primary_cull(rbrt_hip_scene_t* scene, const rbrt_camera_t* cam, uint32_t* out_words, size_t n_words);

/* Kernel timing with HIP events recorded on the launch stream around every trace-kernel launch
 * (and the resolve kernel after it). set_timing(scene, 1) starts / restarts the accumulation;
 * kernel_ms sums the durations of all launches since then (it synchronises on the last event) and
 * returns how many trace launches that was. */
int rbrt_hip_scene_set_timing(rbrt_hip_scene_t* scene, int enable);
int rbrt_hip_scene_kernel_ms(rbrt_hip_scene_t* scene, float* trace_ms_total, float* resolve_ms_total,
                             uint32_t* n_trace_launches);

/* Trace launches since set_timing(scene, 1), by grid: the full grid, or the part of the wave slots a launch of a stream
 * takes (rbrt_hip_scene_set_pipeline; "half" is round 2's name for it), so the mix depends on host timing. */
int rbrt_hip_scene_launch_mix(rbrt_hip_scene_t* scene, uint32_t* n_full_grid, uint32_t* n_half_grid);

/* Helper launches since set_timing(scene, 1): more waves given to launches already running when the GPU has room and the
 * caller has stopped issuing (api.cpp "Elastic launches"; lab knob RBRT_HELPERS=0|1|2: never, automatic, with every launch). */
int rbrt_hip_scene_helper_launches(rbrt_hip_scene_t* scene, uint32_t* n);

/* Where the wall-clock time of rbrt_hip_scene_create (and, for the one-shot rbrt_hip_render, of the whole call) went, in
 * seconds. The parts of create_s: hip_init_s + upload_s + bvh_build_s + lanes_s (+ a remainder of validation and small
 * allocations); of total_s: create_s + render_s + copy_s + destroy_s. bench.py's one_shot leg and the CLI's --report read it. */
typedef struct rbrt_hip_call_times {
    double hip_init_s;  /* device selection: the HIP runtime's own start-up when this is a process's first HIP call */
    double upload_s;    /* the caller's scene arrays to the device */
    double bvh_build_s; /* BVH construction: the host builder's CPU time + upload of its tree, or the device builder's kernels */
    double lanes_s;     /* pipeline lanes (streams, events, per-wave scratch) and loading the device code */
    double create_s;    /* all of rbrt_hip_scene_create */
    double render_s;    /* rbrt_hip_render only: issue of the render to the synchronised device image */
    double copy_s;      /* rbrt_hip_render only: image to the caller's host buffers */
    double destroy_s;   /* rbrt_hip_render only: scene and buffers released */
    double total_s;     /* rbrt_hip_render only */
    uint32_t meshes_device_built, meshes_host_built;
} rbrt_hip_call_times_t;
int rbrt_hip_scene_create_times(rbrt_hip_scene_t* scene, rbrt_hip_call_times_t* out);
int rbrt_hip_last_render_times(rbrt_hip_call_times_t* out); /* of the calling thread's last rbrt_hip_render */

/* A scene handle starts with the trees that cost rbrt_hip_scene_create least -- the device builder's for all but small meshes
 * -- and a background thread makes the host builder's (better) trees of those meshes, which the first render call that
 * finds them ready adopts (api.cpp "The tree a scene STARTS with"; $RBRT_BVH_REFINE=0 turns it off). This waits up to
 * timeout_s for that thread and adopts its result now. *state: 0 = none was started, 1 = its trees are in use, 2 = it
 * failed or was cancelled (rbrt_hip_last_error says why; the first trees stay), 3 = still at work. *build_seconds: start of
 * the thread to its trees on the device. Either pointer may be NULL. Tests and bench.py (whose counting pass and timed
 * legs have to run on ONE tree) use it. */
int rbrt_hip_scene_refine_wait(rbrt_hip_scene_t* scene, double timeout_s, int* state, double* build_seconds);

#ifdef __cplusplus
}
#endif
#endif /* RBRT_HIP_DEBUG_H */
