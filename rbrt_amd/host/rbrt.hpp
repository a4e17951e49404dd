// rbrt.hpp — C++ host-side mirror of the reference's public interface around the hot path.
//
// The reference is a Rust crate; this image has no Rust toolchain, so the host that sits above
// the C ABI (include/rbrt_hip.h) is restated in C++ with the reference's names and argument
// meaning, so a user of rbrt_lib finds the same pieces:
//
//   rbrt_lib::vec3::Vec3                         -> rbrt::Vec3                 (vec3.rs)
//   rbrt_lib::cam::Camera::new                   -> rbrt::Camera::create      (cam.rs:22-62)
//   rbrt_lib::blueprints::*Blueprint             -> rbrt::*Blueprint          (blueprints.rs:15-48)
//   load_blueprints_from_yaml_file               -> same name                 (blueprints.rs:76-92)
//   create_scene_from_scene_blueprint            -> same name                 (blueprints.rs:132-158)
//   rbrt_lib::mesh::TriangleMesh::new            -> rbrt::TriangleMesh::create (mesh.rs:41-74)
//   rbrt_lib::render_scene(cam, samples, scene)  -> rbrt::render_scene        (lib.rs:75-79)
//   image::ImageBuffer<Rgb<u8>>::save            -> rbrt::ImageBuffer::save   (src/main.rs:86)
//
// Where the reference panics (bad YAML, unreadable .obj, unsaveable image) these throw
// rbrt::Error; the CLI turns that into a message and a non-zero exit code.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rbrt_hip.h"

namespace rbrt {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- vec3.rs -----------------------------------------------------------------------------------
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vec3 zero() { return Vec3(); }
    float length() const { return std::sqrt(x * x + y * y + z * z); }
    float sum() const { return x + y + z; }
    Vec3 normalize() const {
        float len = length();
        return Vec3(x / len, y / len, z / len);
    }
    Vec3 cross_product(const Vec3& o) const {
        return Vec3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
    }
    float dot(const Vec3& o) const { return Vec3(x * o.x, y * o.y, z * o.z).sum(); }
    Vec3 rotate_point(const Vec3& rot) const;  // Z-X-Z Euler, vec3.rs:139-155
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vec3 operator*(float s, Vec3 b) { return Vec3(s * b.x, s * b.y, s * b.z); }
inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// ---- cam.rs ------------------------------------------------------------------------------------
struct Camera {
    float hor_fov_rad, vert_fov_rad;
    uint32_t img_width_pix, img_height_pix;
    float img_width_mm, img_height_mm, focal_len_mm;
    Vec3 position, look_at, up, right, img_center_point;
    float mm_per_pix_hor, mm_per_pix_vert;
    // Camera::new(position, look_at, up, img_height_pix, img_width_pix, focal_len_mm)
    static Camera create(Vec3 position, Vec3 look_at, Vec3 up, uint32_t img_height_pix, uint32_t img_width_pix,
                         float focal_len_mm);
    rbrt_camera_t to_abi() const;
};

// ---- materials (blueprints.rs:50-74) -------------------------------------------------------------
struct Material {
    rbrt_material_t abi{};
    static Material lambertian(Vec3 albedo);
    static Material metal(Vec3 albedo, float roughness);
    static Material dielectric(float ref_idx);
};
// None when the type string matches none of metal / lambert / dielectric (the object is dropped).
std::optional<Material> create_material_from_description(const std::string& mat_type, std::optional<Vec3> albedo,
                                                         std::optional<float> material_param);

// ---- blueprints.rs:15-48 -------------------------------------------------------------------------
struct TriangleMeshBlueprint {
    std::string obj_filepath;
    float scale = 1.0f;
    Vec3 translation, rotation_rad;
    std::string material_type;
    std::optional<Vec3> albedo;
    std::optional<float> material_param;
};
struct SphereBlueprint {
    float radius = 0;
    Vec3 center;
    std::string material_type;
    std::optional<Vec3> albedo;
    std::optional<float> material_param;
};
struct CameraBluePrint {
    Vec3 camera_up, camera_look_at, camera_position;
    float camera_focal_length_mm = 0;
};
struct SceneBlueprint {
    CameraBluePrint camera_blueprint;
    std::vector<TriangleMeshBlueprint> mesh_blueprints;
    std::vector<SphereBlueprint> sphere_blueprints;
};
SceneBlueprint load_blueprints_from_yaml_file(const std::string& filepath);
SceneBlueprint load_blueprints_from_yaml_text(const std::string& text);

// ---- sphere.rs / mesh.rs / scene.rs ---------------------------------------------------------------
struct Sphere {
    Vec3 center;
    float radius = 0;
    Material material;
};

// triangle.rs:9-34: a single triangle as a scene element (the reference's second Intersectable). The YAML factory
// never creates one; a C++ caller may (Scene::basic_triangles / element_order below).
struct BasicTriangle {
    std::array<Vec3, 3> corners;  // counter-clockwise
    Material material;
};

struct TriangleMesh {
    // mesh.rs:12-25; vertices[1], vertices[2] are kept like the reference keeps them, although
    // nothing on the hot path reads them.
    std::vector<float> vertices[3][3];
    std::vector<float> edges[2][3];
    std::vector<float> normals[3];
    std::vector<uint8_t> is_padding_triangle;
    Vec3 bbox_lower, bbox_upper;
    Material material;
    uint32_t num_triangles = 0;  // before padding
    // TriangleMesh::new(filepath, translation, rotation, scale, material), mesh.rs:41-74
    static TriangleMesh create(const std::string& filepath, Vec3 translation, Vec3 rotation, float scale,
                               Material material);
    static TriangleMesh from_triangles(std::vector<std::array<Vec3, 3>> pre_vertices, Material material);
    rbrt_mesh_t to_abi() const;
};
constexpr uint32_t kNumVectorLanes = 8;  // mesh.rs:28-30: the AVX layout is the one restated

// tobj::load_obj with default LoadOptions, reduced to what mesh.rs:92-114 consumes:
// per-face vertex triples in file order, then scale -> rotate -> translate (mesh.rs:102-112).
std::vector<std::array<Vec3, 3>> load_mesh_vertices_from_file(const std::string& filepath, Vec3 translation,
                                                              Vec3 rotation, float scale);
Vec3 get_triangle_normal(const std::array<Vec3, 3>& corners);                             // triangle.rs:30-34
void compute_min_max_3d(const std::vector<std::array<Vec3, 3>>& tris, Vec3& lo, Vec3& hi);  // aabbox.rs:62-88

struct Scene {
    // scene.rs:12-16: `elements: Vec<Box<dyn Intersectable + Sync>>` holds Spheres and BasicTriangles. Here the two
    // kinds sit in their own vectors; `element_order` gives the order Scene::hit tests them in (scene.rs:23-31: an
    // earlier element keeps a tie) as entries (index, or 0x80000000 | index for a triangle) -- empty = all spheres,
    // then all triangles, which is what every scene the YAML factory can build looks like.
    std::vector<Sphere> elements;
    std::vector<BasicTriangle> basic_triangles;
    std::vector<uint32_t> element_order;
    std::vector<TriangleMesh> triangle_meshes;
    // POD view for the C ABI; valid while this Scene is alive and unmodified.
    struct AbiView {
        std::vector<rbrt_sphere_t> spheres;
        std::vector<rbrt_triangle_t> triangles;
        std::vector<rbrt_mesh_t> meshes;
        rbrt_scene_t scene{};
    };
    AbiView to_abi() const;
};
Scene create_scene_from_scene_blueprint(const SceneBlueprint& bp);
// Where create_scene_from_scene_blueprint spent its time, summed over the meshes of the calling thread's scenes (--report).
struct LoadTimes {
    double obj_load_s = 0;   // reading and parsing the .obj files, transforming the vertices (mesh.rs:78-121)
    double soa_prep_s = 0;   // edges, normals, padding, SoA arrays, bounding box (mesh.rs:41-74, :123-181)
};
LoadTimes& load_times();

// ---- image + render --------------------------------------------------------------------------------
struct ImageBuffer {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;        // row-major, 3 bytes per pixel
    std::vector<float> radiance;     // row-major fp32 pre-gamma mean (extra to the reference)
    void save(const std::string& path) const;  // .png (8-bit RGB) or .ppm by extension
};
void write_png(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height);

// What render_scene measured (wall clock; with several GPUs the slowest rank's figure). The reference prints only
// "Starting rendering..." and a progress line (lib.rs:80,105-110); a 4 ms render needs more than that to be understood.
struct RenderReport {
    double upload_build_s = 0;   // rbrt_hip_scene_create + the render's buffers, the slowest rank's (= the four below + buffers_s)
    double hip_init_s = 0;       // the HIP runtime's start-up (the process's first HIP call) + device selection
    double upload_s = 0;         // scene arrays to the device
    double bvh_build_s = 0;      // BVH construction (whichever builder made the first trees)
    double lanes_s = 0;          // the library's streams, events, per-wave scratch, device code
    double buffers_s = 0;        // this host's stream and image buffers
    double release_s = 0;        // buffers and scene released
    double render_s = 0;         // all passes, incl. checkpoint writes
    double gather_s = 0;         // image to host memory (and the merge of the ranks' tiles)
    uint32_t passes = 0, pass_spp = 0, checkpoints_written = 0, resumed_from_sample = 0;
    int n_gpus = 1;
    std::string gather = "none";    // none (one GPU) | host | rccl
    std::string builder = "none";   // who built the BVHs: host | device | mixed | none (no mesh)
    uint64_t bvh_nodes = 0, bvh_triangles = 0;
};

struct RenderConfig {  // additions that the reference hard-codes or lacks
    uint64_t seed = 1;
    int n_gpus = 1;
    bool quiet = false;
    uint32_t pass_spp = 0;            // samples per pass (progress line / checkpoint granularity); 0 = automatic
    std::string checkpoint_path;      // non-empty: resume from / write per-pass checkpoints of the running sums here
    int checkpoint_every = 1;         // ... after every n-th pass
    // multi-GPU: "host" (every rank copies its tiles over its own PCIe link, merged on the host: the image has to
    // end up there anyway) or "rccl" (device-to-device over xGMI to GPU 0, then one copy; librccl is dlopen'ed)
    std::string gather = "host";
    bool oversubscribe = false;       // rank r runs on device r % n_devices (rehearsal of N ranks on fewer GPUs; host gather only)
    RenderReport* report = nullptr;   // filled in when not null
};
// rbrt_lib::render_scene (lib.rs:75-79): blocks until the image is complete. Runs on the GPU(s)
// through the C ABI; there is no CPU path.
ImageBuffer render_scene(const Camera& cam, uint32_t num_samples, const Scene& scene, const RenderConfig& cfg = {});

}  // namespace rbrt
