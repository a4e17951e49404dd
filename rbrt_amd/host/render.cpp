// render.cpp — rbrt::render_scene: the call the reference makes at src/main.rs:82, routed through the
// C ABI to the GPU(s). One worker thread per GPU renders its interleaved 8x8 pixel tiles
// (rbrt_render_opts_t::tile_rank / tile_world); partial images are merged on the host.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <thread>

#include "rbrt.hpp"

namespace rbrt {

ImageBuffer render_scene(const Camera& cam, uint32_t num_samples, const Scene& scene, const RenderConfig& cfg) {
    if (!cfg.quiet) std::printf("Starting rendering...\n");
    const rbrt_camera_t c = cam.to_abi();
    const Scene::AbiView view = scene.to_abi();
    rbrt_render_opts_t opts;
    rbrt_render_opts_default(&opts);
    opts.spp = num_samples;
    opts.seed = cfg.seed;

    ImageBuffer img;
    img.width = cam.img_width_pix;
    img.height = cam.img_height_pix;
    const size_t n = size_t(img.width) * img.height * 3;
    img.rgb.assign(n, 0);
    img.radiance.assign(n, 0.0f);

    const int n_dev = rbrt_hip_device_count();
    if (n_dev < 1) throw Error(std::string("no HIP device: ") + rbrt_hip_last_error());
    int world = cfg.n_gpus < 1 ? 1 : cfg.n_gpus;
    if (world > n_dev) throw Error("requested " + std::to_string(world) + " GPUs, " + std::to_string(n_dev) + " present");

    std::vector<std::string> errors(world);
    auto worker = [&](int rank) {
        rbrt_hip_scene_t* hs = nullptr;
        float* d_rad = nullptr;
        uint8_t* d_rgb = nullptr;
        auto fail = [&](const std::string& m) { errors[rank] = m.empty() ? "unknown error" : m; };
        if (rbrt_hip_scene_create(&view.scene, rank, &hs) != RBRT_OK) return fail(rbrt_hip_last_error());
        rbrt_render_opts_t o = opts;
        o.tile_rank = uint32_t(rank);
        o.tile_world = uint32_t(world);
        const size_t npix = world > 1 ? rbrt_hip_packed_pixels(img.width, img.height, o.tile_rank, o.tile_world)
                                      : size_t(img.width) * img.height;
        bool ok = npix == 0 || (hipMalloc(reinterpret_cast<void**>(&d_rad), npix * 3 * sizeof(float)) == hipSuccess &&
                                hipMalloc(reinterpret_cast<void**>(&d_rgb), npix * 3) == hipSuccess);
        if (!ok) fail("hipMalloc failed for the output image");
        if (ok && npix) {
            if (rbrt_hip_render_device(hs, &c, &o, nullptr, d_rad, d_rgb) != RBRT_OK) {
                fail(rbrt_hip_last_error());
            } else if (hipDeviceSynchronize() != hipSuccess) {
                fail("kernel execution failed");
            } else if (world == 1) {
                (void)hipMemcpy(img.radiance.data(), d_rad, n * sizeof(float), hipMemcpyDeviceToHost);
                (void)hipMemcpy(img.rgb.data(), d_rgb, n, hipMemcpyDeviceToHost);
            } else {
                std::vector<float> hr(npix * 3);
                std::vector<uint8_t> hb(npix * 3);
                (void)hipMemcpy(hr.data(), d_rad, hr.size() * sizeof(float), hipMemcpyDeviceToHost);
                (void)hipMemcpy(hb.data(), d_rgb, hb.size(), hipMemcpyDeviceToHost);
                const uint32_t tiles_x = (img.width + RBRT_TILE - 1) / RBRT_TILE;
                for (size_t tl = 0; tl < npix / 64; ++tl) {
                    const uint32_t tile = uint32_t(tl) * world + rank;
                    const uint32_t ty = tile / tiles_x, tx = tile % tiles_x;
                    for (uint32_t p = 0; p < 64; ++p) {
                        const uint32_t row = ty * RBRT_TILE + p / 8, col = tx * RBRT_TILE + p % 8;
                        if (row >= img.height || col >= img.width) continue;
                        const size_t src = (tl * 64 + p) * 3, dst = (size_t(row) * img.width + col) * 3;
                        for (int k = 0; k < 3; ++k) img.radiance[dst + k] = hr[src + k], img.rgb[dst + k] = hb[src + k];
                    }
                }
            }
        }
        if (d_rad) (void)hipFree(d_rad);
        if (d_rgb) (void)hipFree(d_rgb);
        rbrt_hip_scene_destroy(hs);
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < world; ++r) threads.emplace_back(worker, r);
    worker(0);
    for (auto& t : threads) t.join();
    for (int r = 0; r < world; ++r)
        if (!errors[r].empty()) throw Error("GPU " + std::to_string(r) + ": " + errors[r]);
    if (!cfg.quiet) std::printf("\rRendering 100%% complete!\n");
    return img;
}

}  // namespace rbrt
