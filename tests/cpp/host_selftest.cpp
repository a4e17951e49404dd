// host_selftest.cpp — the CPU-side C++ (hand-rolled YAML / .obj parsers, PNG writer, scene assembly, the threaded BVH
// builder) exercised in one process so that it can run under AddressSanitizer + UBSan and under ThreadSanitizer
// (`make -C tests/cpp asan tsan`; GPU sanitizers are not available on this pool, and this code needs no GPU).
// Inputs: the two shipped scenes with a generated .obj, a set of malformed YAML / .obj texts that must be rejected
// with rbrt::Error (the reference panics there: blueprints.rs:80,87, mesh.rs:89) and never crash, and meshes large
// enough to take the builder's multi-threaded path (bvh.cpp splice).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <set>
#include <string>
#include <vector>

#include <atomic>
#include <chrono>
#include <thread>
#include <cstring>
#include "../../rbrt_amd/csrc/bvh.h"
#include "../../rbrt_amd/host/rbrt.hpp"

static int g_failed = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failed;                                                    \
        }                                                                  \
    } while (0)

static std::string slurp(const std::string& p) {
    std::ifstream in(p);
    return std::string((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

static void write_obj(const std::string& path, int n_tris, unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(-0.1f, 0.1f);
    std::ofstream o(path);
    for (int i = 0; i < n_tris; ++i) {
        const float cx = u(rng), cy = u(rng), cz = u(rng);
        for (int k = 0; k < 3; ++k) o << "v " << cx + 0.01f * u(rng) << " " << cy + 0.01f * u(rng) << " " << cz + 0.01f * u(rng) << "\n";
    }
    for (int i = 0; i < n_tris; ++i) {
        if (i % 3 == 0)
            o << "f " << 3 * i + 1 << " " << 3 * i + 2 << " " << 3 * i + 3 << "\n";
        else if (i % 3 == 1)
            o << "f " << 3 * i + 1 << "/1/1 " << 3 * i + 2 << "//2 " << 3 * i + 3 << "/3\n";
        else
            o << "f " << -(3 * (n_tris - i)) << " " << -(3 * (n_tris - i)) + 1 << " " << -(3 * (n_tris - i)) + 2 << "\n";
    }
}

template <class F>
static bool throws(F&& f) {
    try {
        f();
    } catch (const rbrt::Error&) {
        return true;
    } catch (const std::exception&) {
        return true;
    }
    return false;
}

static void check_bvh(const rbrt_mesh_t& m) {
    rbrt::BvhBuildResult r = rbrt::build_bvh(m);
    std::vector<int> seen(m.n_total, 0);
    size_t leaves = 0;
    for (const rbrt::BvhNode4& nd : r.nodes)
        for (int c = 0; c < 4; ++c) {
            const int32_t ch = nd.child[c];
            if (ch == rbrt::kNoChild || ch >= 0) continue;
            const uint32_t leaf = uint32_t(~ch), first = leaf >> rbrt::kLeafBits, cnt = (leaf & uint32_t(rbrt::kLeafMax - 1)) + 1u;
            CHECK(first + cnt <= r.tris.size());
            ++leaves;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t idx = r.tris[first + k].index;
                if (idx != 0xFFFFFFFFu) {
                    CHECK(idx < m.n_total);
                    if (idx < m.n_total) ++seen[idx];
                }
            }
        }
    const uint32_t n_tested = (m.n_total / 8u) * 8u;
    // (a triangle cut by spatial splits is referenced from several leaves; one that the scan cannot return, from none)
    for (uint32_t i = 0; i < m.n_total; ++i) CHECK((i < n_tested && !m.is_padding[i]) ? seen[i] >= 1 : seen[i] == 0);
    CHECK(r.tris.size() <= rbrt::bvh_record_capacity(m.n_total));
    CHECK(r.max_depth <= uint32_t(rbrt::kMaxBvhDepth));
    CHECK(leaves == r.n_leaves || r.n_indexed <= uint32_t(rbrt::kLeafMax));
}

int main(int argc, char** argv) {
    const std::string root = argc > 1 ? argv[1] : ".";
    const std::string tmp = argc > 2 ? argv[2] : "/tmp";
    const std::string obj = tmp + "/selftest_bunny.obj";
    write_obj(obj, 2003, 1);
    // ---- the shipped scenes, through the YAML parser, the material factory, the .obj loader, SoA conversion ----
    for (const char* name : {"scenes/example_scene.yaml", "scenes/header_card.yaml"}) {
        std::string text = slurp(root + "/" + name);
        CHECK(!text.empty());
        size_t p;
        while ((p = text.find("bunny.obj")) != std::string::npos) text.replace(p, 9, obj.substr(0, obj.size() - 4) + "_X.obj");
        while ((p = text.find("_X.obj")) != std::string::npos) text.replace(p, 6, ".obj");
        rbrt::SceneBlueprint bp = rbrt::load_blueprints_from_yaml_text(text);
        rbrt::Scene sc = rbrt::create_scene_from_scene_blueprint(bp);
        CHECK(sc.triangle_meshes.size() == 1 && sc.triangle_meshes[0].num_triangles == 2003);
        CHECK(sc.elements.size() >= 4);
        rbrt::Camera cam = rbrt::Camera::create(bp.camera_blueprint.camera_position, bp.camera_blueprint.camera_look_at,
                                                bp.camera_blueprint.camera_up, 48, 64, bp.camera_blueprint.camera_focal_length_mm);
        CHECK(cam.to_abi().img_width_pix == 64);
        const rbrt::Scene::AbiView view = sc.to_abi();
        check_bvh(view.scene.meshes[0]);
    }
    // ---- malformed YAML: an error, never a crash or an out-of-bounds read ----
    const std::string good = slurp(root + "/scenes/example_scene.yaml");
    std::vector<std::string> bad = {"", "\n\n", "camera_blueprint:", "camera_blueprint:\n  camera_position:\n    x: a\n", "sphere_blueprints: [",
                                    "mesh_blueprints: []\nsphere_blueprints: []\n", "- - -\n", ":\n:\n", "camera_blueprint: {x: 1",
                                    std::string(5000, ' ') + "x", std::string("\t\tcamera_blueprint:\n"), "a: [1, 2, 3\nb: ]"};
    for (size_t cut : {10ul, 57ul, 200ul, 333ul, 700ul, 1200ul, 1900ul}) bad.push_back(good.substr(0, std::min(cut, good.size())));
    {   // single-character corruptions of the good text
        std::mt19937 rng(7);
        for (int k = 0; k < 300; ++k) {
            std::string t = good;
            const size_t pos = rng() % t.size();
            t[pos] = " :-[]{}#x\n\t0"[rng() % 12];
            bad.push_back(t);
        }
    }
    int rejected = 0;
    for (const std::string& t : bad) {
        try {
            rbrt::SceneBlueprint bp = rbrt::load_blueprints_from_yaml_text(t);
            (void)bp;  // some corruptions are still valid scenes: fine
        } catch (const std::exception&) {
            ++rejected;
        }
    }
    CHECK(rejected >= 20);
    // ---- malformed .obj ----
    const char* bad_objs[] = {"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", "v 0 0 0\nf 1 1\n", "v a b c\nv 0 0 0\nv 1 1 1\nf 1 2 3\n",
                              "f 1 2 3\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -4 -1 -2\n",
                              "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/ 2/ 3/\n", "v 1e999 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"};
    int n_bad_obj = 0;
    for (const char* t : bad_objs) {
        const std::string p = tmp + "/selftest_bad.obj";
        std::ofstream(p) << t;
        if (throws([&] { (void)rbrt::load_mesh_vertices_from_file(p, rbrt::Vec3(), rbrt::Vec3(), 1.0f); })) ++n_bad_obj;
    }
    CHECK(n_bad_obj >= 5);
    CHECK(throws([&] { (void)rbrt::load_mesh_vertices_from_file(tmp + "/does_not_exist.obj", rbrt::Vec3(), rbrt::Vec3(), 1.0f); }));
    // ---- PNG ----
    {
        std::vector<uint8_t> rgb(33 * 17 * 3);
        for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = uint8_t(i * 7);
        rbrt::write_png(tmp + "/selftest.png", rgb.data(), 33, 17);
        CHECK(slurp(tmp + "/selftest.png").size() > 60);
        CHECK(throws([&] { rbrt::write_png("/nonexistent_dir/x.png", rgb.data(), 33, 17); }));
    }
    // ---- every image writer (src/main.rs:86: encoder by extension), odd sizes: row padding, run lengths ----
    {
        rbrt::ImageBuffer img;
        img.width = 31, img.height = 7;
        img.rgb.resize(size_t(img.width) * img.height * 3);
        for (size_t i = 0; i < img.rgb.size(); ++i) img.rgb[i] = uint8_t(i < 200 ? 17 : i * 13);
        for (const char* ext : {"png", "ppm", "pnm", "pam", "bmp", "tga", "tif", "tiff", "qoi"}) {
            const std::string p = tmp + "/selftest_img." + ext;
            img.save(p);
            CHECK(slurp(p).size() > 40);
        }
        CHECK(throws([&] { img.save(tmp + "/selftest_img.jpg"); }));
        CHECK(throws([&] { img.save(tmp + "/selftest_img"); }));
        CHECK(throws([&] { img.save("/nonexistent_dir/x.bmp"); }));
        rbrt::ImageBuffer one;
        one.width = one.height = 1;
        one.rgb = {1, 2, 3};
        one.save(tmp + "/selftest_one.qoi");
        one.save(tmp + "/selftest_one.bmp");
    }
    // ---- BasicTriangle elements and the element order in the ABI view (triangle.rs:9-34, scene.rs:23-31) ----
    {
        rbrt::Scene sc;
        sc.elements.push_back(rbrt::Sphere{rbrt::Vec3(0, 0, -5), 1.0f, rbrt::Material::lambertian(rbrt::Vec3(0.5f, 0.5f, 0.5f))});
        sc.basic_triangles.push_back(rbrt::BasicTriangle{{rbrt::Vec3(0, 0, -3), rbrt::Vec3(1, 0, -3), rbrt::Vec3(0, 1, -3)},
                                                         rbrt::Material::metal(rbrt::Vec3(1, 1, 1), 0.1f)});
        sc.element_order = {0x80000000u, 0u};
        const rbrt::Scene::AbiView v = sc.to_abi();
        CHECK(v.scene.n_spheres == 1 && v.scene.n_triangles == 1 && v.scene.element_order && v.scene.element_order[0] == 0x80000000u);
        CHECK(v.scene.triangles[0].corners[1][0] == 1.0f && v.scene.triangles[0].mat.kind == RBRT_MAT_METAL);
        sc.element_order = {0u};
        CHECK(throws([&] { (void)sc.to_abi(); }));
    }
    // ---- the threaded BVH build (>= 32768 triangles): splice of worker subtrees ----
    {
        write_obj(tmp + "/selftest_big.obj", 70003, 3);
        rbrt::TriangleMesh tm = rbrt::TriangleMesh::create(tmp + "/selftest_big.obj", rbrt::Vec3(1, 2, 3), rbrt::Vec3(0.1f, 0.2f, 0.3f), 45.0f,
                                                          rbrt::Material::lambertian(rbrt::Vec3(0.5f, 0.5f, 0.5f)));
        const rbrt_mesh_t m = tm.to_abi();
        check_bvh(m);
        // coincident triangles: no spatial split exists
        std::vector<std::array<rbrt::Vec3, 3>> same(5000, {rbrt::Vec3(0, 0, -5), rbrt::Vec3(1, 0, -5), rbrt::Vec3(0, 1, -5)});
        rbrt::TriangleMesh tm2 = rbrt::TriangleMesh::from_triangles(same, rbrt::Material::metal(rbrt::Vec3(1, 1, 1), 0.1f));
        check_bvh(tm2.to_abi());
        // ---- a build from records in another order (what a scene handle's background thread runs), and cancelling it ----
        const rbrt::BvhBuildResult ref = rbrt::build_bvh(m);
        std::vector<rbrt::BvhTri> recs;  // (one record per triangle, in another order: what the device builder hands over)
        {
            std::vector<char> have(m.n_total, 0);
            for (auto it = ref.tris.rbegin(); it != ref.tris.rend(); ++it)
                if (it->index < m.n_total && !have[it->index]) have[it->index] = 1, recs.push_back(*it);
        }
        const rbrt::BvhBuildResult again = rbrt::build_bvh_from_records(recs.data(), recs.size());
        CHECK(again.nodes.size() == ref.nodes.size() && again.tris.size() == ref.tris.size());
        CHECK(!std::memcmp(again.nodes.data(), ref.nodes.data(), ref.nodes.size() * sizeof(rbrt::BvhNode4)));
        CHECK(!std::memcmp(again.tris.data(), ref.tris.data(), ref.tris.size() * sizeof(rbrt::BvhTri)));
        // raised at every stage of the build (before it, inside the single-threaded top, among the workers, after them):
        // the build returns, flagged, without touching anything out of bounds; an unraised flag changes nothing
        for (int delay_us : {0, 50, 200, 1000, 3000, 8000, 20000, 1000000}) {
            std::atomic<bool> cancel{false};
            rbrt::BvhBuildOptions opt;
            opt.cancel = &cancel;
            std::thread raiser([&] {
                std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
                if (delay_us < 1000000) cancel.store(true);
            });
            if (delay_us == 1000000) raiser.join();  // (never raised)
            const rbrt::BvhBuildResult r = rbrt::build_bvh_from_records(recs.data(), recs.size(), opt);
            if (raiser.joinable()) raiser.join();
            if (!r.cancelled) {
                CHECK(r.nodes.size() == ref.nodes.size() && !std::memcmp(r.nodes.data(), ref.nodes.data(), ref.nodes.size() * sizeof(rbrt::BvhNode4)));
            }
            CHECK(delay_us != 0 || r.cancelled);
            CHECK(delay_us != 1000000 || !r.cancelled);
        }
    }
    if (g_failed) {
        std::fprintf(stderr, "host_selftest: %d check(s) failed\n", g_failed);
        return 1;
    }
    std::printf("host_selftest ok\n");
    return 0;
}
