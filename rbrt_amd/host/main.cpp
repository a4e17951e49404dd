// main.cpp — the `rbrt` command line, same flags and defaults as the reference (src/main.rs:10-50):
//   -t/--target_file dbg_out.png   --height 600   -w/--width 800
//   -c/--config scenes/example_scene.yaml   -s/--samples 5   -h/--help   -V/--version
// plus, not in the reference: --seed N (default 1), --gpus N (default 1), --gather host|rccl, --oversubscribe,
// --pass-samples N, --checkpoint FILE, --checkpoint-every N, --report FILE (machine-readable timing of the run).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rbrt.hpp"

namespace {

void usage() {
    std::printf(
        "a lighweight raytracer written in rust\n\n"
        "Usage: rbrt [OPTIONS]\n\n"
        "Options:\n"
        "  -t, --target_file <target_file>  file that will be created witht he rendered output [default: dbg_out.png]\n"
        "      --height <height>            target image resolution height [default: 600]\n"
        "  -w, --width <width>              target image resolution width [default: 800]\n"
        "  -c, --config <config>            YAML file that specifies the scene layout and camera specification. "
        "[default: scenes/example_scene.yaml]\n"
        "  -s, --samples <samples>          number of rays per pixel [default: 5]\n"
        "      --seed <seed>                seed of the per-(pixel,sample) random streams [default: 1]\n"
        "      --gpus <gpus>                number of MI355X GPUs to shard pixel tiles over [default: 1]\n"
        "      --gather <how>               multi-GPU image gather: host (each GPU over its own PCIe link) or rccl (GPU to GPU\n"
        "                                   over xGMI, then one copy) [default: host]\n"
        "      --oversubscribe              allow more ranks than GPUs (rank r on GPU r mod n; rehearsal, host gather only)\n"
        "      --report <file>              write a JSON timing report of the run there (- for stdout)\n"
        "      --pass-samples <n>           samples per pass (a progress line, and a checkpoint, per pass) [default: automatic]\n"
        "      --checkpoint <file>          write the running per-pixel sums there after passes and resume from it\n"
        "      --checkpoint-every <n>       checkpoint after every n-th pass [default: 1]\n"
        "  -h, --help                       Print help\n"
        "  -V, --version                    Print version\n");
}

uint32_t samples_arg_for_report(uint32_t s) { return s ? s : 1u; }

std::string json_escape(const std::string& in) {
    std::string out;
    for (unsigned char ch : in) {
        if (ch == '"' || ch == '\\') {
            out += '\\', out += char(ch);
        } else if (ch < 0x20) {
            char b[8];
            std::snprintf(b, sizeof(b), "\\u%04x", ch);
            out += b;
        } else {
            out += char(ch);
        }
    }
    return out;
}

bool parse_u32(const char* s, uint32_t& out) {
    char* end = nullptr;
    unsigned long v = std::strtoul(s, &end, 10);
    if (end == s || *end != '\0' || v > 0xFFFFFFFFul || s[0] == '-') return false;
    out = uint32_t(v);
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    // The library overlaps up to eight trace launches on streams of its own; the HIP runtime maps streams onto four hardware
    // queues unless told otherwise, and lanes that share a queue wait for each other (depth_for).
    // Set before the first HIP call; a value the user exported is left alone.
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    std::string target = "dbg_out.png", config = "scenes/example_scene.yaml";
    uint32_t height = 600, width = 800, samples = 5, gpus = 1, pass_samples = 0, checkpoint_every = 1;
    std::string gather = "host", checkpoint, report_path;
    unsigned long long seed = 1;
    bool oversubscribe = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        std::string val;
        bool has_val = false;
        size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) {
            val = a.substr(eq + 1);
            a = a.substr(0, eq);
            has_val = true;
        }
        auto value = [&]() -> const char* {
            if (has_val) return val.c_str();
            if (i + 1 >= argc) {
                std::fprintf(stderr, "error: a value is required for '%s' but none was supplied\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        auto u32 = [&](uint32_t& dst) {
            const char* v = value();
            if (!parse_u32(v, dst)) {
                std::fprintf(stderr, "error: invalid value '%s' for '%s'\n", v, a.c_str());
                std::exit(2);
            }
        };
        if (a == "-h" || a == "--help") {
            usage();
            return 0;
        } else if (a == "-V" || a == "--version") {
            std::printf("rbrt 0.1\n");
            return 0;
        } else if (a == "-t" || a == "--target_file") {
            target = value();
        } else if (a == "--height") {
            u32(height);
        } else if (a == "-w" || a == "--width") {
            u32(width);
        } else if (a == "-c" || a == "--config") {
            config = value();
        } else if (a == "-s" || a == "--samples") {
            u32(samples);
        } else if (a == "--gpus") {
            u32(gpus);
        } else if (a == "--pass-samples") {
            u32(pass_samples);
        } else if (a == "--checkpoint-every") {
            u32(checkpoint_every);
        } else if (a == "--checkpoint") {
            checkpoint = value();
        } else if (a == "--gather") {
            gather = value();
            if (gather != "rccl" && gather != "host") {
                std::fprintf(stderr, "error: invalid value '%s' for '--gather' [possible values: rccl, host]\n", gather.c_str());
                return 2;
            }
        } else if (a == "--oversubscribe") {
            oversubscribe = true;
        } else if (a == "--report") {
            report_path = value();
        } else if (a == "--seed") {
            seed = std::strtoull(value(), nullptr, 10);
        } else {
            std::fprintf(stderr, "error: unexpected argument '%s' found\n\nFor more information, try '--help'.\n",
                         a.c_str());
            return 2;
        }
    }
    using clock = std::chrono::steady_clock;
    const auto secs = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    try {
        const auto t0 = clock::now();
        rbrt::SceneBlueprint bp = rbrt::load_blueprints_from_yaml_file(config);
        rbrt::Camera cam = rbrt::Camera::create(bp.camera_blueprint.camera_position, bp.camera_blueprint.camera_look_at,
                                                bp.camera_blueprint.camera_up, height, width,
                                                bp.camera_blueprint.camera_focal_length_mm);
        const auto t1 = clock::now();
        rbrt::Scene scene = rbrt::create_scene_from_scene_blueprint(bp);  // .obj parse, transform, SoA conversion (mesh.rs:41-181)
        const auto t2 = clock::now();
        rbrt::RenderConfig cfg;
        rbrt::RenderReport rep;
        cfg.report = &rep;
        cfg.oversubscribe = oversubscribe;
        cfg.seed = seed;
        cfg.n_gpus = int(gpus);
        cfg.pass_spp = pass_samples;
        cfg.checkpoint_path = checkpoint;
        cfg.checkpoint_every = int(checkpoint_every);
        cfg.gather = gather;
        rbrt::ImageBuffer img = rbrt::render_scene(cam, samples, scene, cfg);
        std::printf("Saving rendered image to %s\n", target.c_str());
        const auto t3 = clock::now();
        img.save(target);
        const auto t4 = clock::now();
        if (!report_path.empty()) {
            uint64_t triangles = 0;
            for (const auto& m : scene.triangle_meshes) triangles += m.num_triangles;
            const uint32_t spp = samples_arg_for_report(samples);
            const double rendered = double(width) * double(height) * double(spp - rep.resumed_from_sample);  // path samples of THIS run
            std::string js = "{";
            const auto str = [&](const char* k, const std::string& v) { js += std::string("\"") + k + "\": \"" + json_escape(v) + "\", "; };
            const auto num = [&](const char* k, double v, const char* fmt = "%.6f") {
                char b[64];
                std::snprintf(b, sizeof(b), fmt, v);
                js += std::string("\"") + k + "\": " + b + ", ";
            };
            str("config", config), str("target_file", target);
            num("width", width, "%.0f"), num("height", height, "%.0f"), num("samples", samples, "%.0f"), num("seed", double(seed), "%.0f");
            num("gpus", rep.n_gpus, "%.0f"), str("gather", rep.gather);
            num("spheres", double(scene.elements.size()), "%.0f"), num("meshes", double(scene.triangle_meshes.size()), "%.0f");
            num("triangles", double(triangles), "%.0f");
            str("bvh_builder", rep.builder), num("bvh_nodes", double(rep.bvh_nodes), "%.0f"), num("bvh_triangles", double(rep.bvh_triangles), "%.0f");
            num("passes", rep.passes, "%.0f"), num("pass_samples", rep.pass_spp, "%.0f");
            num("checkpoints_written", rep.checkpoints_written, "%.0f"), num("resumed_from_sample", rep.resumed_from_sample, "%.0f");
            // where the run's time went; the parts add up to total_s (other_s is what none of them covers: thread start,
            // checkpoint look-up, the report itself)
            const rbrt::LoadTimes lt = rbrt::load_times();
            const double parse_s = secs(t0, t1), prep_s = std::max(0.0, secs(t1, t2) - lt.obj_load_s), encode_s = secs(t3, t4), total_s = secs(t0, t4);
            const double named = parse_s + lt.obj_load_s + prep_s + rep.hip_init_s + rep.upload_s + rep.bvh_build_s + rep.lanes_s + rep.buffers_s +
                                 rep.render_s + rep.gather_s + rep.release_s + encode_s;
            num("parse_s", parse_s), num("obj_load_s", lt.obj_load_s), num("prep_s", prep_s), num("hip_init_s", rep.hip_init_s);
            num("upload_s", rep.upload_s), num("bvh_build_s", rep.bvh_build_s), num("lanes_s", rep.lanes_s), num("buffers_s", rep.buffers_s);
            num("render_s", rep.render_s), num("gather_s", rep.gather_s), num("release_s", rep.release_s), num("encode_s", encode_s);
            num("other_s", std::max(0.0, total_s - named)), num("total_s", total_s);
            num("upload_build_s", rep.upload_build_s);  // (= hip_init_s' library part + upload_s + bvh_build_s + lanes_s + buffers_s: the name earlier reports used)
            num("mray_samples_per_s", rep.render_s > 0 ? rendered / rep.render_s / 1e6 : 0.0, "%.3f");
            js.resize(js.size() - 2);
            js += "}\n";
            if (report_path == "-") {
                std::fputs(js.c_str(), stdout);
            } else {
                FILE* f = std::fopen(report_path.c_str(), "w");
                if (!f) throw rbrt::Error("cannot write report " + report_path);
                std::fputs(js.c_str(), f);
                std::fclose(f);
            }
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rbrt: %s\n", e.what());
        return 101;  // the exit code of a Rust panic
    }
    return 0;
}
