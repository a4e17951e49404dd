// scene.cpp — cold path of the reference restated for the C++ host: Vec3::rotate_point, Camera::new,
// the YAML blueprints, the material factory, the .obj loader and the SoA mesh conversion with its
// padding rule. Arithmetic keeps the reference's f32 evaluation order: what is computed here is
// the exact triangle set and camera the kernel sees.
#include <chrono>
#include <algorithm>
#include <array>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

#include "rbrt.hpp"
#include "yaml_lite.hpp"

namespace rbrt {

// vec3.rs:139-155
Vec3 Vec3::rotate_point(const Vec3& rot) const {
    float s_x = std::sin(rot.x), s_y = std::sin(rot.y), s_z = std::sin(rot.z);
    float c_x = std::cos(rot.x), c_y = std::cos(rot.y), c_z = std::cos(rot.z);
    return Vec3((c_x * c_z - c_y * s_x * s_z) * x - (c_x * s_z + c_y * c_z * s_x) * y + s_x * s_y * z,
                (c_z * s_x + c_x * c_y * s_z) * x + (c_x * c_y * c_z - s_x * s_z) * y - c_x * s_y * z,
                s_y * s_z * x + c_z * s_y * y + c_y * z);
}

// cam.rs:22-62
Camera Camera::create(Vec3 position, Vec3 look_at, Vec3 up, uint32_t img_height_pix, uint32_t img_width_pix,
                      float focal_len_mm) {
    Camera c;
    c.right = look_at.normalize().cross_product(up.normalize()).normalize();
    c.img_width_mm = 35.0f;  // full frame sensor
    c.mm_per_pix_hor = c.img_width_mm / float(img_width_pix);
    c.img_height_mm = float(img_height_pix) * c.mm_per_pix_hor;
    c.mm_per_pix_vert = c.img_height_mm / float(img_height_pix);
    c.img_center_point = position + focal_len_mm / 1000.0f * look_at.normalize();
    c.hor_fov_rad = 2.0f * std::atan(2.0f * focal_len_mm / c.img_width_mm);
    c.vert_fov_rad = 2.0f * std::atan(2.0f * focal_len_mm / c.img_height_mm);
    c.img_width_pix = img_width_pix;
    c.img_height_pix = img_height_pix;
    c.position = position;
    c.focal_len_mm = focal_len_mm;
    c.up = up;
    c.look_at = look_at;
    return c;
}

static void put3(float* dst, Vec3 v) { dst[0] = v.x, dst[1] = v.y, dst[2] = v.z; }

rbrt_camera_t Camera::to_abi() const {
    rbrt_camera_t a{};
    put3(a.position, position);
    put3(a.right, right);
    put3(a.up, up);
    put3(a.img_center_point, img_center_point);
    a.mm_per_pix_hor = mm_per_pix_hor;
    a.mm_per_pix_vert = mm_per_pix_vert;
    a.img_width_pix = img_width_pix;
    a.img_height_pix = img_height_pix;
    return a;
}

// ---- materials -----------------------------------------------------------------------------------
Material Material::lambertian(Vec3 albedo) {
    Material m;
    m.abi.kind = RBRT_MAT_LAMBERTIAN;
    put3(m.abi.albedo, albedo);
    return m;
}
Material Material::metal(Vec3 albedo, float roughness) {
    Material m;
    m.abi.kind = RBRT_MAT_METAL;
    put3(m.abi.albedo, albedo);
    m.abi.param = roughness;
    return m;
}
Material Material::dielectric(float ref_idx) {
    Material m;
    m.abi.kind = RBRT_MAT_DIELECTRIC;
    m.abi.param = ref_idx;
    return m;
}

// blueprints.rs:50-74: substring match on the lower-cased type, in the order metal, lambert, dielectric.
std::optional<Material> create_material_from_description(const std::string& mat_type, std::optional<Vec3> albedo,
                                                         std::optional<float> material_param) {
    std::string t = mat_type;
    std::transform(t.begin(), t.end(), t.begin(), [](unsigned char c) { return char(std::tolower(c)); });
    if (t.find("metal") != std::string::npos) {
        if (!albedo) throw Error("you forgot to specify an albedo vector for metal");
        if (!material_param) throw Error("you forgot to specify a roughness (i.e. material_param: 0.1) for metal");
        return Material::metal(*albedo, *material_param);
    }
    if (t.find("lambert") != std::string::npos) {
        if (!albedo) throw Error("you forgot to specify an albedo vector for lambertian");
        return Material::lambertian(*albedo);
    }
    if (t.find("dielectric") != std::string::npos) {
        if (!material_param)
            throw Error("you forgot to specify a refractory index vector (i.e. material_param: 1.8) dielectric");
        return Material::dielectric(*material_param);
    }
    std::printf("Cannot figure out material_type from %s, material_type must be one of metal, lambertian or dielectric!\n",
                mat_type.c_str());
    return std::nullopt;
}

// ---- YAML -> blueprints ----------------------------------------------------------------------------
namespace {

using yaml_lite::Node;

const Node& need(const Node& m, const char* key, const char* where) {
    if (m.kind != Node::Map) throw Error(std::string(where) + ": expected a mapping");
    const Node* n = m.find(key);
    if (!n) throw Error(std::string(where) + ": missing field `" + key + "`");
    return *n;
}

// serde_yaml hands a YAML float to an f32 field as f64 -> `as f32` (two roundings); mirrored here.
float as_f32(const Node& n, const char* what) {
    if (n.kind != Node::Scalar || n.quoted || n.scalar.empty()) throw Error(std::string(what) + ": expected a number");
    std::string s = n.scalar;
    if (s == ".inf" || s == ".Inf" || s == ".INF" || s == "+.inf") return std::numeric_limits<float>::infinity();
    if (s == "-.inf" || s == "-.Inf" || s == "-.INF") return -std::numeric_limits<float>::infinity();
    if (s == ".nan" || s == ".NaN" || s == ".NAN") return std::numeric_limits<float>::quiet_NaN();
    s.erase(std::remove(s.begin(), s.end(), '_'), s.end());
    char* end = nullptr;
    double d = std::strtod(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') throw Error(std::string(what) + ": invalid number `" + n.scalar + "`");
    return float(d);
}

Vec3 as_vec3(const Node& n, const char* what) {
    return Vec3(as_f32(need(n, "x", what), what), as_f32(need(n, "y", what), what), as_f32(need(n, "z", what), what));
}

std::string as_string(const Node& n, const char* what) {
    if (n.kind != Node::Scalar) throw Error(std::string(what) + ": expected a string");
    return n.scalar;
}

std::optional<Vec3> opt_vec3(const Node& m, const char* key, const char* what) {
    const Node* n = m.find(key);
    if (!n || n->is_null()) return std::nullopt;
    return as_vec3(*n, what);
}
std::optional<float> opt_f32(const Node& m, const char* key, const char* what) {
    const Node* n = m.find(key);
    if (!n || n->is_null()) return std::nullopt;
    return as_f32(*n, what);
}

const std::vector<Node>& as_list(const Node& n, const char* what) {
    static const std::vector<Node> empty;
    if (n.kind == Node::List) return n.list;
    throw Error(std::string(what) + ": expected a sequence");
}

}  // namespace

SceneBlueprint load_blueprints_from_yaml_text(const std::string& text) {
    Node root;
    try {
        root = yaml_lite::parse(text);
    } catch (const std::exception& e) {
        throw Error(std::string("Unable to parse content to scene blueprint: ") + e.what());
    }
    SceneBlueprint bp;
    const Node& cam = need(root, "camera_blueprint", "scene");
    bp.camera_blueprint.camera_up = as_vec3(need(cam, "camera_up", "camera_blueprint"), "camera_up");
    bp.camera_blueprint.camera_look_at = as_vec3(need(cam, "camera_look_at", "camera_blueprint"), "camera_look_at");
    bp.camera_blueprint.camera_position = as_vec3(need(cam, "camera_position", "camera_blueprint"), "camera_position");
    bp.camera_blueprint.camera_focal_length_mm =
        as_f32(need(cam, "camera_focal_length_mm", "camera_blueprint"), "camera_focal_length_mm");
    for (const Node& m : as_list(need(root, "mesh_blueprints", "scene"), "mesh_blueprints")) {
        TriangleMeshBlueprint b;
        b.obj_filepath = as_string(need(m, "obj_filepath", "mesh blueprint"), "obj_filepath");
        b.scale = as_f32(need(m, "scale", "mesh blueprint"), "scale");
        b.translation = as_vec3(need(m, "translation", "mesh blueprint"), "translation");
        b.rotation_rad = as_vec3(need(m, "rotation_rad", "mesh blueprint"), "rotation_rad");
        b.material_type = as_string(need(m, "material_type", "mesh blueprint"), "material_type");
        b.albedo = opt_vec3(m, "albedo", "albedo");
        b.material_param = opt_f32(m, "material_param", "material_param");
        bp.mesh_blueprints.push_back(b);
    }
    for (const Node& s : as_list(need(root, "sphere_blueprints", "scene"), "sphere_blueprints")) {
        SphereBlueprint b;
        b.radius = as_f32(need(s, "radius", "sphere blueprint"), "radius");
        b.center = as_vec3(need(s, "center", "sphere blueprint"), "center");
        b.material_type = as_string(need(s, "material_type", "sphere blueprint"), "material_type");
        b.albedo = opt_vec3(s, "albedo", "albedo");
        b.material_param = opt_f32(s, "material_param", "material_param");
        bp.sphere_blueprints.push_back(b);
    }
    return bp;
}

SceneBlueprint load_blueprints_from_yaml_file(const std::string& filepath) {
    std::ifstream f(filepath, std::ios::binary);
    if (!f) throw Error("Failed to open \"" + filepath + "\" to load content.");
    std::stringstream ss;
    ss << f.rdbuf();
    try {
        return load_blueprints_from_yaml_text(ss.str());
    } catch (const Error& e) {
        throw Error("Unable to parse content of file \"" + filepath + "\" to scene blueprint: " + e.what());
    }
}

// ---- .obj ---------------------------------------------------------------------------------------
namespace {

// One "model" in tobj's sense: a run of faces between o / g / usemtl statements. Its flat index list
// is cut into triples (mesh.rs:96-113 assumes triangles; with tobj's default triangulate=false a
// polygon's indices simply continue the list, and that is reproduced).
struct ObjModel {
    std::vector<uint32_t> indices;
};

bool parse_index(const char*& p, long n_vertices, uint32_t& out) {
    char* end = nullptr;
    long v = std::strtol(p, &end, 10);
    if (end == p) return false;
    p = end;
    while (*p && *p != ' ' && *p != '\t') ++p;  // skip /vt/vn
    long idx = v > 0 ? v - 1 : n_vertices + v;    // 1-based, negative = relative to the end
    if (v == 0 || idx < 0 || idx >= n_vertices) return false;
    out = uint32_t(idx);
    return true;
}

}  // namespace

std::vector<std::array<Vec3, 3>> load_mesh_vertices_from_file(const std::string& filepath, Vec3 translation,
                                                              Vec3 rotation, float scale) {
    std::ifstream f(filepath, std::ios::binary);
    if (!f) throw Error("assertion failed: loaded_mesh.is_ok() (cannot open " + filepath + ")");
    std::vector<float> positions;
    std::vector<ObjModel> models(1);
    std::string line;
    size_t line_no = 0;
    while (std::getline(f, line)) {
        ++line_no;
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            float v[3];
            for (int k = 0; k < 3; ++k) {
                char* end = nullptr;
                v[k] = std::strtof(p, &end);  // Rust's str::parse::<f32>: correctly rounded
                if (end == p) throw Error(filepath + ":" + std::to_string(line_no) + ": bad vertex");
                p = end;
            }
            positions.insert(positions.end(), v, v + 3);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            const long nv = long(positions.size() / 3);
            for (;;) {
                while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
                if (!*p) break;
                uint32_t idx;
                if (!parse_index(p, nv, idx)) throw Error(filepath + ":" + std::to_string(line_no) + ": bad face index");
                models.back().indices.push_back(idx);
            }
        } else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t' || p[1] == '\0' || p[1] == '\r')) {
            if (!models.back().indices.empty()) models.emplace_back();
        } else if (std::strncmp(p, "usemtl", 6) == 0) {
            if (!models.back().indices.empty()) models.emplace_back();
        }
    }
    std::vector<std::array<Vec3, 3>> model_vertices;
    for (const ObjModel& m : models) {
        for (size_t fidx = 0; fidx < m.indices.size() / 3; ++fidx) {
            std::array<Vec3, 3> tri;
            for (int k = 0; k < 3; ++k) {
                const uint32_t i = m.indices[3 * fidx + k];
                Vec3 scaled(positions[3 * i] * scale, positions[3 * i + 1] * scale, positions[3 * i + 2] * scale);
                tri[k] = scaled.rotate_point(rotation) + translation;  // mesh.rs:102-112
            }
            model_vertices.push_back(tri);
        }
    }
    std::printf("Successfully loaded %zu triangles from file %s!\n", model_vertices.size(), filepath.c_str());
    return model_vertices;
}

Vec3 get_triangle_normal(const std::array<Vec3, 3>& c) {
    Vec3 edge1 = c[1] - c[0], edge2 = c[2] - c[0];
    return edge1.cross_product(edge2).normalize();
}

void compute_min_max_3d(const std::vector<std::array<Vec3, 3>>& tris, Vec3& lo, Vec3& hi) {
    const float fmax = std::numeric_limits<float>::max();
    lo = Vec3(fmax, fmax, fmax);
    hi = Vec3(-fmax, -fmax, -fmax);
    for (const auto& tri : tris)
        for (const Vec3& v : tri) {
            if (v.x < lo.x) lo.x = v.x;
            if (v.y < lo.y) lo.y = v.y;
            if (v.z < lo.z) lo.z = v.z;
            if (v.x > hi.x) hi.x = v.x;
            if (v.y > hi.y) hi.y = v.y;
            if (v.z > hi.z) hi.z = v.z;
        }
}

// mesh.rs:41-74 + convert_to_soa_mesh (mesh.rs:123-181) including its padding rule: N % lanes copies
// of triangle 0 are appended (not lanes - N % lanes); the kernel side then drops the incomplete chunk.
TriangleMesh TriangleMesh::from_triangles(std::vector<std::array<Vec3, 3>> pre_vertices, Material material) {
    TriangleMesh m;
    m.material = material;
    m.num_triangles = uint32_t(pre_vertices.size());
    std::vector<Vec3> pre_normals;
    std::vector<std::array<Vec3, 2>> pre_edges;
    for (const auto& t : pre_vertices) pre_normals.push_back(get_triangle_normal(t));
    for (const auto& t : pre_vertices) pre_edges.push_back({t[1] - t[0], t[2] - t[0]});
    compute_min_max_3d(pre_vertices, m.bbox_lower, m.bbox_upper);
    // mesh.rs:27-38 determine_num_vector_lanes(): the reference announces the SIMD layout it picked, once per mesh,
    // after the "Successfully loaded" line. This host always lays meshes out 8 lanes wide (the AVX layout is the one
    // the GPU path restates); on a CPU without AVX the line says so instead of claiming a capability.
    if (__builtin_cpu_supports("avx"))
        std::printf("AVX capability detected!\n");
    else
        std::printf("AVX capability not detected - the GPU path uses the 8-lane (AVX) mesh layout regardless!\n");
    const size_t n = pre_vertices.size();
    const size_t n_pad = n % kNumVectorLanes;
    m.is_padding_triangle.assign(n, 0);
    for (size_t i = 0; i < n_pad; ++i) {
        pre_normals.push_back(pre_normals[0]);
        pre_edges.push_back(pre_edges[0]);
        pre_vertices.push_back(pre_vertices[0]);
        m.is_padding_triangle.push_back(1);
    }
    for (const auto& t : pre_vertices)
        for (int k = 0; k < 3; ++k) {
            m.vertices[k][0].push_back(t[k].x);
            m.vertices[k][1].push_back(t[k].y);
            m.vertices[k][2].push_back(t[k].z);
        }
    for (const auto& e : pre_edges)
        for (int k = 0; k < 2; ++k) {
            m.edges[k][0].push_back(e[k].x);
            m.edges[k][1].push_back(e[k].y);
            m.edges[k][2].push_back(e[k].z);
        }
    for (const Vec3& nn : pre_normals) {
        m.normals[0].push_back(nn.x);
        m.normals[1].push_back(nn.y);
        m.normals[2].push_back(nn.z);
    }
    return m;
}

LoadTimes& load_times() {
    thread_local LoadTimes t;
    return t;
}

TriangleMesh TriangleMesh::create(const std::string& filepath, Vec3 translation, Vec3 rotation, float scale,
                                  Material material) {
    const auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    auto tris = load_mesh_vertices_from_file(filepath, translation, rotation, scale);
    const auto t1 = now();
    TriangleMesh m = from_triangles(std::move(tris), material);
    load_times().obj_load_s += std::chrono::duration<double>(t1 - t0).count();
    load_times().soa_prep_s += std::chrono::duration<double>(now() - t1).count();
    return m;
}

rbrt_mesh_t TriangleMesh::to_abi() const {
    rbrt_mesh_t a{};
    a.n_total = uint32_t(is_padding_triangle.size());
    a.n_real = num_triangles;
    a.v0x = vertices[0][0].data(), a.v0y = vertices[0][1].data(), a.v0z = vertices[0][2].data();
    a.e1x = edges[0][0].data(), a.e1y = edges[0][1].data(), a.e1z = edges[0][2].data();
    a.e2x = edges[1][0].data(), a.e2y = edges[1][1].data(), a.e2z = edges[1][2].data();
    a.nx = normals[0].data(), a.ny = normals[1].data(), a.nz = normals[2].data();
    a.is_padding = is_padding_triangle.data();
    put3(a.bbox_lo, bbox_lower);
    put3(a.bbox_hi, bbox_upper);
    a.mat = material.abi;
    return a;
}

Scene::AbiView Scene::to_abi() const {
    AbiView v;
    for (const Sphere& s : elements) {
        rbrt_sphere_t a{};
        put3(a.center, s.center);
        a.radius = s.radius;
        a.mat = s.material.abi;
        v.spheres.push_back(a);
    }
    for (const BasicTriangle& t : basic_triangles) {
        rbrt_triangle_t a{};
        for (int k = 0; k < 3; ++k) put3(a.corners[k], t.corners[k]);
        a.mat = t.material.abi;
        v.triangles.push_back(a);
    }
    for (const TriangleMesh& m : triangle_meshes) v.meshes.push_back(m.to_abi());
    v.scene.n_spheres = uint32_t(v.spheres.size());
    v.scene.spheres = v.spheres.data();
    v.scene.n_triangles = uint32_t(v.triangles.size());
    v.scene.triangles = v.triangles.data();
    if (!element_order.empty()) {
        if (element_order.size() != elements.size() + basic_triangles.size())
            throw Error("Scene::element_order must name every sphere and triangle exactly once");
        v.scene.element_order = element_order.data();
    }
    v.scene.n_meshes = uint32_t(v.meshes.size());
    v.scene.meshes = v.meshes.data();
    return v;
}

// blueprints.rs:132-158: meshes first (loading the files), then spheres; objects whose material
// cannot be built are dropped.
Scene create_scene_from_scene_blueprint(const SceneBlueprint& bp) {
    Scene scene;
    for (const TriangleMeshBlueprint& mb : bp.mesh_blueprints) {
        auto mat = create_material_from_description(mb.material_type, mb.albedo, mb.material_param);
        if (!mat) {
            std::printf("Failed to parse material info provided with mesh!\n");
            continue;
        }
        scene.triangle_meshes.push_back(
            TriangleMesh::create(mb.obj_filepath, mb.translation, mb.rotation_rad, mb.scale, *mat));
    }
    for (const SphereBlueprint& sb : bp.sphere_blueprints) {
        auto mat = create_material_from_description(sb.material_type, sb.albedo, sb.material_param);
        if (!mat) continue;
        scene.elements.push_back(Sphere{sb.center, sb.radius, *mat});
    }
    return scene;
}

}  // namespace rbrt
