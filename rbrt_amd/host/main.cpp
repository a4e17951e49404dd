// main.cpp — the `rbrt` command line, same flags and defaults as the reference (src/main.rs:10-50):
//   -t/--target_file dbg_out.png   --height 600   -w/--width 800
//   -c/--config scenes/example_scene.yaml   -s/--samples 5   -h/--help   -V/--version
// plus, not in the reference: --seed N (default 1), --gpus N (default 1), --gather rccl|host, --pass-samples N,
// --checkpoint FILE, --checkpoint-every N.
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
        "      --gather <how>               multi-GPU image gather: rccl (GPU to GPU over xGMI) or host [default: rccl]\n"
        "      --pass-samples <n>           samples per pass (a progress line, and a checkpoint, per pass) [default: automatic]\n"
        "      --checkpoint <file>          write the running per-pixel sums there after passes and resume from it\n"
        "      --checkpoint-every <n>       checkpoint after every n-th pass [default: 1]\n"
        "  -h, --help                       Print help\n"
        "  -V, --version                    Print version\n");
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
    std::string target = "dbg_out.png", config = "scenes/example_scene.yaml";
    uint32_t height = 600, width = 800, samples = 5, gpus = 1, pass_samples = 0, checkpoint_every = 1;
    std::string gather = "rccl", checkpoint;
    unsigned long long seed = 1;
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
        } else if (a == "--seed") {
            seed = std::strtoull(value(), nullptr, 10);
        } else {
            std::fprintf(stderr, "error: unexpected argument '%s' found\n\nFor more information, try '--help'.\n",
                         a.c_str());
            return 2;
        }
    }
    try {
        rbrt::SceneBlueprint bp = rbrt::load_blueprints_from_yaml_file(config);
        rbrt::Camera cam = rbrt::Camera::create(bp.camera_blueprint.camera_position, bp.camera_blueprint.camera_look_at,
                                                bp.camera_blueprint.camera_up, height, width,
                                                bp.camera_blueprint.camera_focal_length_mm);
        rbrt::Scene scene = rbrt::create_scene_from_scene_blueprint(bp);
        rbrt::RenderConfig cfg;
        cfg.seed = seed;
        cfg.n_gpus = int(gpus);
        cfg.pass_spp = pass_samples;
        cfg.checkpoint_path = checkpoint;
        cfg.checkpoint_every = int(checkpoint_every);
        cfg.gather = gather;
        rbrt::ImageBuffer img = rbrt::render_scene(cam, samples, scene, cfg);
        std::printf("Saving rendered image to %s\n", target.c_str());
        img.save(target);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rbrt: %s\n", e.what());
        return 101;  // the exit code of a Rust panic
    }
    return 0;
}
