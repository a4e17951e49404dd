/*
 * rbrt_hip.h — C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * This is the drop-in boundary for the ONE call the reference makes into its
 * render hot path:
 *
 *     rbrt_lib::render_scene(cam: Camera, num_samples: u32, scene: Scene)
 *         -> image::ImageBuffer<Rgb<u8>, Vec<u8>>          (rbrt_lib/src/lib.rs:75-79,
 *                                                           called at src/main.rs:82)
 *
 * Everything below is plain-old-data: borrowed pointers + sizes, no callbacks,
 * no C++/torch types. A Rust host binds it with an `extern "C"` block (see
 * INTEGRATION.md); the C++ host in rbrt_amd/host/ and the Python ctypes mirror
 * in rbrt_amd/abi.py bind the same symbols.
 *
 * Conventions
 *   - every entry point returns 0 (RBRT_OK) or a negative rbrt_status_t; it never
 *     aborts or throws across the boundary. rbrt_hip_last_error() returns a
 *     thread-local human-readable message for the last failure.
 *   - all pointers are borrowed for the duration of the call only (scene_create
 *     copies what it needs to the device before returning).
 *   - images are row-major, row 0 = TOP row, 3 channels interleaved (RGB):
 *       radiance: float[H][W][3]  — linear, pre-gamma mean over samples (lib.rs:95-101)
 *       rgb8    : uint8[H][W][3]  — (sqrt(c)*256) saturating cast (lib.rs:116-122)
 *   - there is NO CPU fallback: without a usable HIP device every render entry
 *     point fails with RBRT_ERR_NO_DEVICE.
 *   - environment: the library reads four variables, all optional --
 *       RBRT_HIP_WORKSPACE_MB   cap of one pipeline lane's sample workspace in MiB (default 1024)
 *       RBRT_BVH_BUILDER        host | device: force one BVH builder (default: a mesh's first tree by whichever builder
 *                               costs the call less, the host builder's tree following from a background thread)
 *       RBRT_BVH_REFINE         0: no background build, a handle keeps the tree it started with
 *       RBRT_BVH_THREADS        threads of the host BVH builder (default: the machine's, at most 16)
 *     and takes note of GPU_MAX_HW_QUEUES as the process had it when the library was loaded (the HIP runtime's own
 *     variable: the frame pipeline is planned for the hardware queues the runtime really has).
 *   - threads: besides the caller's, a scene handle may own a background BVH builder and, once it has seen a stream of
 *     calls, a watcher that gives launches still running the wave slots that have become free (rbrt_hip_scene_destroy
 *     ends both). A handle is still used by ONE calling thread at a time.
 *     Everything else that tunes the kernels' scheduling is a lab knob: ignored unless RBRT_HIP_LAB=1 is set, and
 *     documented with the test / diagnostic entry points in rbrt_hip_debug.h, not here. No knob changes the image.
 */
#ifndef RBRT_HIP_H
#define RBRT_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBRT_ABI_VERSION 2 /* 2: rbrt_scene_t grew n_triangles / triangles / element_order (appended: the v1 prefix is unchanged);
                              entry points added since (rbrt_hip_tile_xy / _tile_number) change no struct and no existing call */

typedef enum rbrt_status {
    RBRT_OK = 0,
    RBRT_ERR_INVALID_ARG = -1,
    RBRT_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at init */
    RBRT_ERR_HIP = -3,         /* a HIP call failed; see rbrt_hip_last_error() */
    RBRT_ERR_OOM = -4,
    RBRT_ERR_UNSUPPORTED = -5, /* e.g. max_depth above the kernel's compiled limit */
    RBRT_ERR_NAN = -6          /* reference would have panicked: sphere.rs:33 "Encountered NAN" */
} rbrt_status_t;

/* Material = the closed set the reference's YAML factory can build
 * (blueprints.rs:50-74); replaces `Box<dyn RayScattering + Sync>` (materials.rs:4-12). */
typedef enum rbrt_material_kind {
    RBRT_MAT_LAMBERTIAN = 0, /* lambertian.rs:11-24: albedo            */
    RBRT_MAT_METAL = 1,      /* metal.rs:12-25     : albedo, param=roughness */
    RBRT_MAT_DIELECTRIC = 2  /* dielectric.rs:11-59: param=ref_idx     */
} rbrt_material_kind_t;

typedef struct rbrt_material {
    int32_t kind;    /* rbrt_material_kind_t */
    float albedo[3]; /* ignored for dielectric (attenuation is (1,1,1), dielectric.rs:18) */
    float param;     /* metal: roughness; dielectric: ref_idx; lambertian: unused */
} rbrt_material_t;

/* sphere.rs:6-10 */
typedef struct rbrt_sphere {
    float center[3];
    float radius;
    rbrt_material_t mat;
} rbrt_sphere_t;

/* mesh.rs:12-25 as produced by convert_to_soa_mesh (mesh.rs:123-181).
 * Every SoA array has n_total = N + N % 8 entries (the reference's padding rule,
 * mesh.rs:134-144, kept as is). The library applies the chunks_exact(8)
 * truncation of triangle.rs:166-167 itself: only the first 8*floor(n_total/8)
 * entries are ever tested, and entries with is_padding[i] != 0 never win
 * (triangle.rs:400). vertices[1], vertices[2] of the reference struct are never
 * read on the hot path and are not part of the ABI. */
typedef struct rbrt_mesh {
    uint32_t n_total;          /* length of each array below */
    uint32_t n_real;           /* N = triangles loaded from the .obj (informational) */
    const float* v0x;          /* vertices[0][0..2]  (triangle.rs:177-179) */
    const float* v0y;
    const float* v0z;
    const float* e1x;          /* edges[0][0..2] = v1 - v0 (mesh.rs:57-60) */
    const float* e1y;
    const float* e1z;
    const float* e2x;          /* edges[1][0..2] = v2 - v0 */
    const float* e2y;
    const float* e2z;
    const float* nx;           /* normals[0..2] = normalize(e1 x e2) (triangle.rs:30-34) */
    const float* ny;
    const float* nz;
    const uint8_t* is_padding; /* mesh.rs:138-144 */
    float bbox_lo[3];          /* BoundingBox of the N real triangles (mesh.rs:62, aabbox.rs:62-88) */
    float bbox_hi[3];
    rbrt_material_t mat;       /* one material per mesh (mesh.rs:24) */
} rbrt_mesh_t;

/* triangle.rs:9-34: BasicTriangle, the reference's second `Intersectable` (lib.rs:38-41) besides Sphere -- a single
 * triangle as a scene element, corners counter-clockwise. The YAML factory never creates one (blueprints.rs:132-158
 * builds spheres and meshes only), but Scene::elements is a Vec<Box<dyn Intersectable>> and admits it; a host that
 * fills it with triangles gets them rendered. Intersection: triangle.rs:92-130 (scalar Moller-Trumbore, |a| < min_dist
 * rejects, 0 <= u <= 1, v >= 0, u + v <= 1, t > min_dist, then the distance window) and :412-441; the hit normal is
 * normalize((c1 - c0) x (c2 - c0)) (triangle.rs:30-34), computed by the library, never flipped towards the ray. */
typedef struct rbrt_triangle {
    float corners[3][3];
    rbrt_material_t mat;
} rbrt_triangle_t;

/* scene.rs:12-16 (lights are always empty in the reference: blueprints.rs:151) */
typedef struct rbrt_scene {
    uint32_t n_spheres;
    const rbrt_sphere_t* spheres; /* the spheres of Scene::elements, YAML order (scene.rs:23-31) */
    uint32_t n_meshes;
    const rbrt_mesh_t* meshes;    /* Scene::triangle_meshes, YAML order (scene.rs:33-41) */
    uint32_t n_triangles;
    const rbrt_triangle_t* triangles; /* the BasicTriangles of Scene::elements */
    /* Order of Scene::elements (scene.rs:23-31 tests them in that order and the EARLIER element wins a tie in distance):
     * NULL = all spheres, then all triangles; else n_spheres + n_triangles entries, entry k = the k-th element:
     * bit 31 clear -> spheres[entry], bit 31 set -> triangles[entry & 0x7fffffff]; every object exactly once.
     * Object ids (rbrt_hip_trace_rays) follow this order: element k has id k, mesh m has id n_spheres + n_triangles + m. */
    const uint32_t* element_order;
} rbrt_scene_t;

/* The 8 of Camera's 14 fields (cam.rs:4-19) that get_ray_through_pixel (cam.rs:64-82) reads. */
typedef struct rbrt_camera {
    float position[3];
    float right[3];
    float up[3]; /* raw YAML vector, not normalised (cam.rs:55,75) */
    float img_center_point[3];
    float mm_per_pix_hor;
    float mm_per_pix_vert;
    uint32_t img_width_pix;
    uint32_t img_height_pix;
} rbrt_camera_t;

/* Values the reference hard-codes are fields here with those defaults
 * (rbrt_render_opts_default): max depth 50 (lib.rs:99), min/max dist
 * (lib.rs:44-45), background (lib.rs:89-93). `seed` replaces the reference's
 * OS-seeded rand::random (cam.rs:69,71; materials.rs:17-19; dielectric.rs:48):
 * the random stream of sample s of pixel (row, col) is a pure function of
 * (seed, row*W+col, s), consumed in the reference's draw order. */
typedef struct rbrt_render_opts {
    uint32_t spp;       /* num_samples */
    uint32_t max_depth; /* 50 */
    float min_dist;     /* 0.001 */
    float max_dist;     /* 2000.0 */
    float bg[3];        /* (0.05, 0.05, 0.8) */
    uint64_t seed;
    /* pixel-tile sharding for multi-GPU: 8x8 tiles are dealt round-robin by tile NUMBER ("How tiles are
     * dealt to ranks", below), this call renders the tiles numbered t with t % tile_world == tile_rank.
     * tile_world <= 1 renders the whole image. */
    uint32_t tile_rank;
    uint32_t tile_world;
    uint32_t flags;     /* RBRT_FLAG_* */
    uint32_t reserved;
} rbrt_render_opts_t;

#define RBRT_FLAG_NONE 0u
#define RBRT_FLAG_COLLECT_STATS 1u /* run the counting variant of the kernel (slower); see rbrt_hip_stats_t */

#define RBRT_TILE 8u /* tile edge in pixels used for sharding and work ordering */
/* How tiles are dealt to ranks. Tile NUMBER t (0 <= t < tiles_x * tiles_y) belongs to rank t % tile_world, and a rank's packed
 * buffers hold its tiles in ascending number. Number t is the image tile in tile row ty = t / tiles_x at tile column
 * tx = (t % tiles_x + RBRT_TILE_SKEW * ty) % tiles_x: every tile row is rotated by RBRT_TILE_SKEW more than the one above, so
 * that a rank's tiles lie on skew lines through the image and not in fixed tile columns (with 8 ranks and a width of 128 or
 * 240 tiles a rank owned every 8th COLUMN, and the columns over the mesh made two ranks 5 % slower than the mean at
 * 1920 x 1080). rbrt_hip_unpack_tiles undoes it; rbrt_hip_tile_xy / rbrt_hip_tile_number (below) are the two directions for a
 * host that does its own gather. The image does not depend on the dealing. */
#define RBRT_TILE_SKEW 3u

/* Per-render work counters from the counting kernel variant: the inputs of the
 * algorithmic-bytes figure ("Measurement"). */
typedef struct rbrt_hip_stats {
    uint64_t rays;            /* Scene::hit calls (scene.rs:19) */
    uint64_t mesh_gate_pass;  /* rays x meshes that passed BoundingBox::hit (aabbox.rs:28-58) */
    uint64_t nodes_visited;   /* BVH node records fetched */
    uint64_t tris_tested;     /* Moller-Trumbore evaluations (triangle.rs:189-255 per lane) */
    uint64_t mesh_hits;       /* accepted mesh hits (normal fetch, mesh.rs:253-257) */
    uint64_t samples;         /* paths traced */
    uint64_t nan_discriminants; /* sphere.rs:33 would have panicked */
    uint32_t node_bytes;      /* size of one BVH node record */
    uint32_t tri_bytes;       /* size of one device triangle record */
} rbrt_hip_stats_t;

typedef struct rbrt_hip_scene rbrt_hip_scene_t; /* opaque: device-resident scene + BVH + workspace */

/* ---- one-shot convenience: the closest analogue of lib.rs:75-79 --------------------------- */

/* Renders the whole image (tile_world ignored unless > 1, then only this rank's tiles are
 * written and all other pixels are left untouched). out_radiance and out_rgb8 are HOST buffers of
 * H*W*3 elements; either may be NULL. */
int rbrt_hip_render(const rbrt_camera_t* cam, const rbrt_scene_t* scene,
                    const rbrt_render_opts_t* opts, float* out_radiance, uint8_t* out_rgb8);

/* ---- resident scene: upload + BVH build once, render many times --------------------------- */

int rbrt_hip_scene_create(const rbrt_scene_t* scene, int device, rbrt_hip_scene_t** out);
int rbrt_hip_scene_destroy(rbrt_hip_scene_t* scene);

/* Number of pixels this rank owns for a W x H image under (tile_rank, tile_world), counting
 * the padded pixels of partial edge tiles (each tile contributes RBRT_TILE*RBRT_TILE slots). */
size_t rbrt_hip_packed_pixels(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_world);

/* Tile number -> tile row / column and back ("How tiles are dealt to ranks", above); tiles_x = ceil(width / RBRT_TILE). */
void rbrt_hip_tile_xy(uint32_t tile, uint32_t tiles_x, uint32_t* tile_row, uint32_t* tile_col);
uint32_t rbrt_hip_tile_number(uint32_t tile_row, uint32_t tile_col, uint32_t tiles_x);

/* Renders into DEVICE memory on `stream` (a hipStream_t, may be NULL = default stream) and
 * returns without synchronising.
 * Threading rule: a scene handle is used by ONE host thread at a time and all its render_device calls go to the
 * SAME stream (the handle's accumulator, its pipeline lanes and their events are ordered through that stream);
 * use one handle per stream / thread otherwise (the reference's render_scene is not re-entrant either: it takes
 * Scene by value, lib.rs:75-79).
 *   d_radiance: if tile_world <= 1: float[H][W][3] row-major.
 *               else: this rank's tiles packed, float[n_local_tiles][64][3] (tile-local pixel
 *               p = (y%8)*8 + (x%8)); feed the gathered buffers to rbrt_hip_unpack_tiles.
 *   d_rgb8    : same indexing with uint8 elements; may be NULL. d_radiance may be NULL too. */
int rbrt_hip_render_device(rbrt_hip_scene_t* scene, const rbrt_camera_t* cam,
                           const rbrt_render_opts_t* opts, void* stream, float* d_radiance,
                           uint8_t* d_rgb8);

/* Progressive / checkpointed rendering (no counterpart in the reference, whose sample loop lib.rs:95-101 runs to the
 * end or not at all; a 4096 x 4096 x 4096 spp render is 68.7 G paths). Renders samples [sample_begin, sample_end) of
 * the opts->spp samples of every pixel and adds them, in sample order, to the running sums in d_accum (device,
 * fp32, indexed like d_radiance; read unless sample_begin == 0). The call with sample_end == opts->spp also
 * writes the mean to d_radiance and its quantisation to d_rgb8 (either may be NULL). Calls must cover [0, spp) in
 * ascending, non-overlapping ranges; the final image is then bit-identical to one rbrt_hip_render_device call,
 * because the per-pixel additions happen in the same order. A checkpoint is d_accum plus sample_end. */
int rbrt_hip_render_pass(rbrt_hip_scene_t* scene, const rbrt_camera_t* cam, const rbrt_render_opts_t* opts,
                         void* stream, uint32_t sample_begin, uint32_t sample_end, float* d_accum,
                         float* d_radiance, uint8_t* d_rgb8);

/* De-interleave gathered per-rank packed tile buffers (concatenated rank 0..world-1, each
 * rbrt_hip_packed_pixels(...)*3 floats, device memory) into a row-major float[H][W][3] device
 * image and/or its uint8 quantisation. Runs on `stream`. */
int rbrt_hip_unpack_tiles(int device, void* stream, const float* d_gathered, uint32_t width,
                          uint32_t height, uint32_t tile_world, float* d_radiance, uint8_t* d_rgb8);

/* Same, for gathered buffers laid out in equal-size slots: rank r's tiles start at pixel slot
 * r * rank_stride_pixels (what a gather of equal-size tensors produces; rank_stride_pixels >=
 * rbrt_hip_packed_pixels(w, h, 0, world), the largest share). rank_stride_pixels = 0 means tightly packed. */
int rbrt_hip_unpack_tiles_strided(int device, void* stream, const float* d_gathered, uint32_t width,
                                  uint32_t height, uint32_t tile_world, size_t rank_stride_pixels,
                                  float* d_radiance, uint8_t* d_rgb8);

/* What scene_create built (informational; the C++ host's --report prints it). */
typedef struct rbrt_hip_scene_info {
    uint32_t n_spheres, n_meshes;
    uint32_t n_meshes_device_built; /* meshes whose BVH the GPU builder made (the host's SAH builder made the rest) */
    uint32_t bvh_stack_need;        /* traversal stack entries the deepest tree can need */
    uint64_t n_nodes;               /* 128-B BVH nodes, all meshes */
    uint64_t n_triangles;           /* indexed triangle records, all meshes (what the reference's scan can return) */
    uint32_t trace_waves;           /* resident single-wave workgroups of a full trace launch */
    uint32_t lds_bytes_per_wave;    /* LDS each of them uses */
    uint32_t occupancy_api_waves_per_cu; /* hipOccupancyMaxActiveBlocksPerMultiprocessor for that kernel and LDS size */
    uint32_t n_cus;
} rbrt_hip_scene_info_t;
int rbrt_hip_scene_info(rbrt_hip_scene_t* scene, rbrt_hip_scene_info_t* out);

/* Error state of the resident path. rbrt_hip_render_device returns before the kernels have run, so what they
 * detect cannot come back through its return value: a NaN sphere discriminant (the reference panics with
 * "Encountered NAN", sphere.rs:33; here those rays miss that sphere and are counted) or a corrupt path slot
 * (internal error). This call synchronises the device, returns RBRT_ERR_NAN / RBRT_ERR_HIP if either happened in
 * any render on this scene since the previous check, and clears the flags. The one-shot rbrt_hip_render does the
 * same check itself. Call it where the reference's render_scene would have returned (lib.rs:124). */
int rbrt_hip_scene_check(rbrt_hip_scene_t* scene);

/* Counters of the last render on this scene that had RBRT_FLAG_COLLECT_STATS set. */
int rbrt_hip_scene_stats(rbrt_hip_scene_t* scene, rbrt_hip_stats_t* out);

/* Frame pipeline depth of a scene handle: 1..8, or 0 = automatic (the default, or $RBRT_PIPELINE): 8 in a process that
 * has exported GPU_MAX_HW_QUEUES=8 before the HIP runtime started, else 4 (the runtime's four hardware queues run four
 * launches side by side). A launch issued into a stream of launches -- while another launch of the scene is still
 * running -- takes a part of the GPU's wave slots (3 or 4 of a CU's 16 at depth 8, 6 at depth 4), one that finds the
 * GPU idle takes them all, and the sample batches of one blocking call are sized by the batches behind them (api.cpp
 * grid_for). Lanes beyond the four made by rbrt_hip_scene_create, and every lane's sample buffer, are made when a
 * stream of calls is first seen; a blocking caller uses the lanes there are.
 * With depth d > 1 consecutive
 * trace launches -- the sample batches of one render and successive rbrt_hip_render_device calls -- alternate
 * over d internal streams and d sets of work buffers, so that a launch's last, poorly filled waves overlap
 * with the start of the next launch; the per-pixel resolve (and with it every write to the caller's output
 * buffers) stays on the caller's stream, in call order. No counterpart in the reference (its render_scene
 * is one blocking call, lib.rs:75-124); results are identical for every depth. */
int rbrt_hip_scene_set_pipeline(rbrt_hip_scene_t* scene, uint32_t depth);

/* ---- misc ---------------------------------------------------------------------------------- */

void rbrt_render_opts_default(rbrt_render_opts_t* opts); /* spp = 5 (src/main.rs:47), seed = 1 */
int rbrt_hip_device_count(void);                         /* >= 0, or negative status */
const char* rbrt_hip_last_error(void);
int rbrt_hip_abi_version(void);
#ifdef __cplusplus
}
#endif
#endif /* RBRT_HIP_H */
