// render.cpp — rbrt::render_scene: the call the reference makes at src/main.rs:82, routed through the C ABI to the
// GPU(s).
//
// One worker thread per GPU. Every GPU holds the scene + BVH and renders its interleaved 8x8 pixel tiles
// (rbrt_render_opts_t::tile_rank / tile_world) in PASSES of samples (rbrt_hip_render_pass: per-pixel sums in sample
// order, so the image does not depend on how the samples are cut into passes). After each pass the reference's
// progress line is printed (lib.rs:105-110 prints one per finished column; here a pass is the unit of progress) and,
// if asked for, a checkpoint is written: the running sums of every rank plus the number of samples done. A later run
// with the same scene, size, sample count, seed and GPU count resumes from it.
// When all samples are in, the image has to reach HOST memory (the caller saves a file: src/main.rs:86). Default
// (`--gather host`): every rank copies its own packed tiles over its own PCIe link and de-interleaves them on the host --
// N parallel transfers of 1/N of the image each. `--gather rccl`: the packed fp32 radiance of ranks 1..N-1 goes to rank
// 0's GPU with ONE grouped RCCL send/recv (each peer over its own xGMI link: SURVEY 8(e)), rank 0 de-interleaves it
// with rbrt_hip_unpack_tiles and quantises, then the whole image crosses ONE PCIe link; that is the shape bench.py's
// device-resident gather has, kept here for hosts that want the image on GPU 0. librccl is loaded with dlopen the first
// time that path is asked for: a single-GPU run, or the host gather, does not depend on it.
// RenderConfig::oversubscribe (CLI --oversubscribe) maps rank r to device r % n_devices, so that every line of the
// N-rank path except the RCCL calls themselves (which need distinct devices) runs on a box with fewer GPUs.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is not linked

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <thread>

#include "rbrt.hpp"
#include "../../include/rbrt_hip_debug.h"

namespace rbrt {
namespace {

class Barrier {  // all-ranks rendezvous between passes (checkpoint consistency, RCCL group entry)
  public:
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        const uint64_t gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            ++gen_;
            cv_.notify_all();
        } else {
            cv_.wait(lk, [&] { return gen_ != gen; });
        }
    }

  private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    uint64_t gen_ = 0;
};

uint64_t fnv1a(const void* p, size_t n, uint64_t h) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}

// What a checkpoint must match to be resumed: everything the running sums depend on.
uint64_t scene_fingerprint(const rbrt_camera_t& cam, const rbrt_scene_t& sc) {
    uint64_t h = 0xCBF29CE484222325ull;
    h = fnv1a(&cam, sizeof(cam), h);
    for (uint32_t i = 0; i < sc.n_spheres; ++i) h = fnv1a(&sc.spheres[i], sizeof(rbrt_sphere_t), h);
    for (uint32_t i = 0; i < sc.n_meshes; ++i) {
        const rbrt_mesh_t& m = sc.meshes[i];
        h = fnv1a(&m.n_total, sizeof(m.n_total), h);
        h = fnv1a(&m.mat, sizeof(m.mat), h);
        const float* arrs[12] = {m.v0x, m.v0y, m.v0z, m.e1x, m.e1y, m.e1z, m.e2x, m.e2y, m.e2z, m.nx, m.ny, m.nz};
        for (const float* a : arrs) h = fnv1a(a, size_t(m.n_total) * sizeof(float), h);
        h = fnv1a(m.is_padding, m.n_total, h);
    }
    return h;
}

struct CheckpointHeader {
    char magic[8];  // "RBRTCKP1"
    uint32_t width, height, spp, world;
    uint64_t seed, fingerprint;
    uint32_t samples_done, reserved;
};

// librccl, loaded on first use (`--gather rccl` with more than one GPU).
struct RcclApi {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void* handle = nullptr;

    static const RcclApi& get() {  // throws rbrt::Error when the library or a symbol is missing
        static const RcclApi api = load();
        return api;
    }

  private:
    static RcclApi load() {
        RcclApi a;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.handle) break;
        }
        if (!a.handle) throw Error(std::string("--gather rccl: cannot load librccl.so (") + dlerror() + "); use --gather host");
        auto sym = [&](const char* n) {
            void* p = dlsym(a.handle, n);
            if (!p) throw Error(std::string("--gather rccl: librccl.so lacks ") + n);
            return p;
        };
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(sym("ncclCommAbort"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        return a;
    }
};

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

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
    if (num_samples == 0) throw Error("the number of samples must be at least 1");

    const auto t_hip0 = std::chrono::steady_clock::now();
    const int n_dev = rbrt_hip_device_count();  // (the process's first HIP call: the runtime starts here)
    const double hip_start_s = seconds_since(t_hip0);
    if (n_dev < 1) throw Error(std::string("no HIP device: ") + rbrt_hip_last_error());
    const int world = cfg.n_gpus < 1 ? 1 : cfg.n_gpus;
    if (world > n_dev && !cfg.oversubscribe)
        throw Error("requested " + std::to_string(world) + " GPUs, " + std::to_string(n_dev) + " present");
    const auto device_of = [&](int rank) { return cfg.oversubscribe ? rank % n_dev : rank; };
    // --gather rccl is a request, not a condition: whatever keeps RCCL from doing the gather -- the library is not
    // there, two ranks share a device, the communicators do not come up, the exchange itself fails -- sends the ranks
    // to the host gather in the same process (their tiles are still in their own memories), and the run says so on
    // stderr and in --report. `rccl_note` is the reason, once there is one.
    bool use_rccl = world > 1 && cfg.gather == "rccl";
    std::string rccl_note;
    if (use_rccl && world > n_dev) {
        rccl_note = "--gather rccl needs one GPU per rank (RCCL cannot put two ranks on one device)";
        use_rccl = false;
    }

    // samples per pass: what was asked for, else passes of about 2^31 path samples each (a second or so of one GPU's
    // share), so that long renders report progress and can checkpoint; a short render is one pass
    uint32_t pass_spp = cfg.pass_spp;
    if (pass_spp == 0) {
        const uint64_t per_spp = uint64_t(img.width) * img.height / uint64_t(world) + 1u;
        pass_spp = uint32_t(std::min<uint64_t>(num_samples, std::max<uint64_t>(1, (1ull << 31) / per_spp)));
    }
    const uint32_t ckpt_every = cfg.checkpoint_every < 1 ? 1u : uint32_t(cfg.checkpoint_every);
    // test hook: give up after this many passes of THIS run (as an interrupted run would), checkpoint left on disk
    int stop_after = 0;
    if (const char* e = std::getenv("RBRT_TEST_STOP_AFTER_PASS")) stop_after = std::atoi(e);

    // ---- resume ------------------------------------------------------------------------------------------------
    CheckpointHeader want{};
    std::memcpy(want.magic, "RBRTCKP1", 8);
    want.width = img.width, want.height = img.height, want.spp = num_samples, want.world = uint32_t(world);
    want.seed = cfg.seed;
    want.fingerprint = cfg.checkpoint_path.empty() ? 0 : scene_fingerprint(c, view.scene);
    uint32_t start_sample = 0;
    std::vector<std::vector<float>> resume_acc(world);
    if (!cfg.checkpoint_path.empty()) {
        std::ifstream in(cfg.checkpoint_path, std::ios::binary);
        CheckpointHeader h{};
        if (in && in.read(reinterpret_cast<char*>(&h), sizeof(h))) {
            const bool same = !std::memcmp(h.magic, want.magic, 8) && h.width == want.width && h.height == want.height &&
                              h.spp == want.spp && h.world == want.world && h.seed == want.seed &&
                              h.fingerprint == want.fingerprint && h.samples_done > 0 && h.samples_done < num_samples;
            bool ok = same;
            for (int r = 0; ok && r < world; ++r) {
                uint64_t cnt = 0;
                ok = bool(in.read(reinterpret_cast<char*>(&cnt), sizeof(cnt)));
                const size_t expect = world > 1 ? rbrt_hip_packed_pixels(img.width, img.height, uint32_t(r), uint32_t(world)) * 3 : n;
                ok = ok && cnt == expect;
                if (ok) {
                    resume_acc[r].resize(cnt);
                    ok = bool(in.read(reinterpret_cast<char*>(resume_acc[r].data()), std::streamsize(cnt * sizeof(float))));
                }
            }
            if (ok) {
                start_sample = h.samples_done;
                if (!cfg.quiet) std::printf("Resuming from checkpoint %s at sample %u of %u\n", cfg.checkpoint_path.c_str(), start_sample, num_samples);
            } else {
                for (auto& v : resume_acc) v.clear();
                if (!cfg.quiet) std::printf("Checkpoint %s does not match this render (scene, size, samples, seed or GPU count): starting over\n",
                                            cfg.checkpoint_path.c_str());
            }
        }
    }

    // ---- RCCL communicators (one process, one communicator per GPU) -----------------------------------------------
    const RcclApi* rccl = nullptr;
    std::vector<ncclComm_t> comms(world, nullptr);
    if (use_rccl) {
        try {
            rccl = &RcclApi::get();
            std::vector<int> devs(world);
            for (int r = 0; r < world; ++r) devs[r] = device_of(r);
            const ncclResult_t rc = rccl->CommInitAll(comms.data(), world, devs.data());
            if (rc != ncclSuccess) rccl_note = std::string("ncclCommInitAll: ") + rccl->GetErrorString(rc);
        } catch (const Error& e) {
            rccl_note = e.what();
        }
        if (!rccl_note.empty()) {
            for (ncclComm_t& cm : comms) {
                if (cm && rccl) (void)rccl->CommAbort(cm);
                cm = nullptr;
            }
            use_rccl = false;
        }
    }
    if (!rccl_note.empty()) std::fprintf(stderr, "warning: %s; gathering through host memory instead\n", rccl_note.c_str());

    // Errors: a rank records its first failure under the mutex and raises the counter. Ranks never have to agree on
    // whether to go on rendering (a failed rank keeps meeting the barriers, idle); they DO have to agree on entering
    // the RCCL group, and that decision is read between two barriers, when nobody can be failing.
    std::mutex err_mutex;
    std::vector<std::string> errors(world);
    std::atomic<int> n_failed{0};
    std::atomic<int> rccl_failed{0};  // ranks whose part of the RCCL exchange failed: not a failure of the run (host gather instead)
    std::vector<std::vector<float>> ckpt_acc(world);  // host copies of the running sums for the checkpoint writer
    Barrier barrier(world);
    float* d_slots = nullptr;  // rank 0, RCCL gather: world equal-size slots of packed tiles
    const size_t slot_pixels = world > 1 ? rbrt_hip_packed_pixels(img.width, img.height, 0, uint32_t(world)) : 0;
    RenderReport rep;
    rep.n_gpus = world;
    rep.pass_spp = pass_spp;
    rep.gather = world > 1 ? (use_rccl ? "rccl" : "host") : "none";
    if (!rccl_note.empty()) rep.gather = "host (rccl was asked for: " + rccl_note + ")";
    std::vector<double> t_setup(world, 0.0), t_render(world, 0.0), t_gather(world, 0.0), t_release(world, 0.0);
    std::vector<rbrt_hip_call_times_t> t_create(world);

    auto worker = [&](int rank) {
        rbrt_hip_scene_t* hs = nullptr;
        float *d_acc = nullptr, *d_rad = nullptr, *d_img = nullptr;
        uint8_t* d_rgb = nullptr;
        hipStream_t stream = nullptr;
        bool failed = false;  // this rank's own view: it has recorded an error
        auto fail = [&](const std::string& m) {
            if (failed) return;
            failed = true;
            {
                std::lock_guard<std::mutex> lk(err_mutex);
                errors[rank] = m.empty() ? "unknown error" : m;
            }
            n_failed.fetch_add(1);
        };
        auto hip_ok = [&](hipError_t e, const char* what) {
            if (e != hipSuccess) fail(std::string(what) + ": " + hipGetErrorString(e));
            return e == hipSuccess;
        };
        const auto t_start = std::chrono::steady_clock::now();
        rbrt_render_opts_t o = opts;
        o.tile_rank = uint32_t(rank);
        o.tile_world = uint32_t(world);
        const int dev = device_of(rank);
        const size_t npix = world > 1 ? rbrt_hip_packed_pixels(img.width, img.height, o.tile_rank, o.tile_world)
                                      : size_t(img.width) * img.height;
        if (rbrt_hip_scene_create(&view.scene, dev, &hs) != RBRT_OK) fail(rbrt_hip_last_error());
        std::memset(&t_create[rank], 0, sizeof(t_create[rank]));
        if (hs) (void)rbrt_hip_scene_create_times(hs, &t_create[rank]);
        // (every pass is followed by a synchronisation here: only the sample batches INSIDE a pass overlap, on three lanes;
        // the library's default of eight is for streams of frames)
        if (hs) (void)rbrt_hip_scene_set_pipeline(hs, 3);
        if (hs && rank == 0) {
            rbrt_hip_scene_info_t info;
            if (rbrt_hip_scene_info(hs, &info) == RBRT_OK) {
                rep.bvh_nodes = info.n_nodes, rep.bvh_triangles = info.n_triangles;
                rep.builder = info.n_meshes == 0 ? "none" : info.n_meshes_device_built == info.n_meshes ? "device" : info.n_meshes_device_built == 0 ? "host" : "mixed";
            }
        }
        if (hs && npix) {
            hip_ok(hipSetDevice(dev), "hipSetDevice");
            hip_ok(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
            hip_ok(hipMalloc(reinterpret_cast<void**>(&d_acc), npix * 3 * sizeof(float)), "hipMalloc(sums)");
            hip_ok(hipMalloc(reinterpret_cast<void**>(&d_rad), npix * 3 * sizeof(float)), "hipMalloc(radiance)");
            if (world == 1) hip_ok(hipMalloc(reinterpret_cast<void**>(&d_rgb), npix * 3), "hipMalloc(rgb8)");
            if (rank == 0 && use_rccl) {
                hip_ok(hipMalloc(reinterpret_cast<void**>(&d_slots), size_t(world) * slot_pixels * 3 * sizeof(float)), "hipMalloc(slots)");
                hip_ok(hipMalloc(reinterpret_cast<void**>(&d_img), n * sizeof(float)), "hipMalloc(image)");
                hip_ok(hipMalloc(reinterpret_cast<void**>(&d_rgb), n), "hipMalloc(rgb8)");
            }
            if (!failed && start_sample != 0)
                hip_ok(hipMemcpy(d_acc, resume_acc[rank].data(), npix * 3 * sizeof(float), hipMemcpyHostToDevice), "upload of the checkpoint");
        }
        t_setup[rank] = seconds_since(t_start);
        // ---- passes ---- (every rank meets every barrier, failed or not: the loop bounds are the same for all)
        const auto t_passes = std::chrono::steady_clock::now();
        uint32_t pass_no = 0;
        for (uint64_t b = start_sample; b < num_samples; b += pass_spp, ++pass_no) {
            const uint32_t e = uint32_t(std::min<uint64_t>(num_samples, b + pass_spp));
            const bool last = e == num_samples;
            if (!failed && npix) {
                if (rbrt_hip_render_pass(hs, &c, &o, stream, uint32_t(b), e, d_acc, last ? d_rad : nullptr, last ? (world == 1 ? d_rgb : nullptr) : nullptr) != RBRT_OK)
                    fail(rbrt_hip_last_error());
                else
                    hip_ok(hipStreamSynchronize(stream), "render pass");
            }
            const bool ckpt_now = !cfg.checkpoint_path.empty() && !last && (pass_no + 1) % ckpt_every == 0;
            if (ckpt_now && !failed && npix) {
                ckpt_acc[rank].resize(npix * 3);
                hip_ok(hipMemcpy(ckpt_acc[rank].data(), d_acc, npix * 3 * sizeof(float), hipMemcpyDeviceToHost), "download of the running sums");
            }
            if (world > 1 && (ckpt_now || !cfg.quiet)) barrier.wait();  // every rank has finished the pass
            if (rank == 0) {
                if (!cfg.quiet) {  // lib.rs:105-110
                    std::printf("\rRendering %.1f%% complete!", double(e) / double(num_samples) * 100.0);
                    std::fflush(stdout);
                }
                // (between the two barriers of a checkpointing pass no other rank is running: the counter is stable)
                if (ckpt_now && n_failed.load() == 0) {
                    const std::string tmp = cfg.checkpoint_path + ".tmp";
                    std::ofstream out(tmp, std::ios::binary | std::ios::trunc);
                    CheckpointHeader h = want;
                    h.samples_done = e;
                    out.write(reinterpret_cast<const char*>(&h), sizeof(h));
                    for (int r = 0; r < world; ++r) {
                        const uint64_t cnt = ckpt_acc[r].size();
                        out.write(reinterpret_cast<const char*>(&cnt), sizeof(cnt));
                        out.write(reinterpret_cast<const char*>(ckpt_acc[r].data()), std::streamsize(cnt * sizeof(float)));
                    }
                    out.close();
                    if (!out || std::rename(tmp.c_str(), cfg.checkpoint_path.c_str()) != 0) fail("cannot write checkpoint " + cfg.checkpoint_path);
                    if (!failed) ++rep.checkpoints_written;
                }
            }
            if (world > 1 && ckpt_now) barrier.wait();  // the sums may change again only after they are on disk
            if (stop_after > 0 && int(pass_no) + 1 == stop_after && !last) fail("stopped after pass " + std::to_string(stop_after) + " (RBRT_TEST_STOP_AFTER_PASS)");
        }
        if (rank == 0) rep.passes = pass_no;
        // ---- the reference would have panicked on a NaN discriminant (sphere.rs:33): surface it ----
        if (hs && !failed && rbrt_hip_scene_check(hs) != RBRT_OK) fail(rbrt_hip_last_error());
        t_render[rank] = seconds_since(t_passes);
        // ---- gather ----
        const auto t_g = std::chrono::steady_clock::now();
        // --gather host: every rank's tiles over its own PCIe link, merged on the host (also where a failed RCCL gather ends)
        auto host_gather = [&]() {
            if (failed || !npix) return;
            std::vector<float> hr(npix * 3);
            if (hip_ok(hipMemcpy(hr.data(), d_rad, hr.size() * sizeof(float), hipMemcpyDeviceToHost), "download")) {
                const uint32_t tiles_x = (img.width + RBRT_TILE - 1) / RBRT_TILE;
                for (size_t tl = 0; tl < npix / 64; ++tl) {  // (ranks write disjoint pixels of the shared image)
                    const uint32_t tile = uint32_t(tl) * uint32_t(world) + uint32_t(rank);
                    uint32_t ty, tx;
                    rbrt_hip_tile_xy(tile, tiles_x, &ty, &tx);
                    for (uint32_t p = 0; p < 64; ++p) {
                        const uint32_t row = ty * RBRT_TILE + p / 8, col = tx * RBRT_TILE + p % 8;
                        if (row >= img.height || col >= img.width) continue;
                        const size_t src = (tl * 64 + p) * 3, dst = (size_t(row) * img.width + col) * 3;
                        for (int k = 0; k < 3; ++k) {  // lib.rs:116-122 on the host for this path
                            const float v = hr[src + k];
                            img.radiance[dst + k] = v;
                            const float q = std::sqrt(v) * 256.0f;
                            img.rgb[dst + k] = !(q == q) || q <= 0.0f ? 0 : q >= 255.0f ? 255 : uint8_t(q);
                        }
                    }
                }
            }
        };

        if (world == 1) {
            if (!failed && npix) {
                hip_ok(hipMemcpy(img.radiance.data(), d_rad, n * sizeof(float), hipMemcpyDeviceToHost), "download");
                hip_ok(hipMemcpy(img.rgb.data(), d_rgb, n, hipMemcpyDeviceToHost), "download");
            }
        } else if (use_rccl) {
            // The decision to enter the group is taken ONCE, between two barriers: before the first every rank has
            // recorded what it had to record, and until the second nobody runs code that can fail. All enter or none.
            barrier.wait();
            const bool go = n_failed.load() == 0;
            barrier.wait();
            if (go) {
                // one grouped exchange: rank r > 0 sends its packed tiles, rank 0 receives each into that rank's slot
                bool copied = true;
                if (rank == 0) copied = hip_ok(hipMemcpyAsync(d_slots, d_rad, npix * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream), "own tiles");
                ncclResult_t rc = rccl->GroupStart();
                if (rank == 0) {
                    for (int r = 1; r < world && rc == ncclSuccess; ++r) {
                        const size_t cnt = rbrt_hip_packed_pixels(img.width, img.height, uint32_t(r), uint32_t(world)) * 3;
                        if (cnt) rc = rccl->Recv(d_slots + size_t(r) * slot_pixels * 3, cnt, ncclFloat, r, comms[0], stream);
                    }
                } else if (npix && rc == ncclSuccess) {
                    rc = rccl->Send(d_rad, npix * 3, ncclFloat, 0, comms[rank], stream);
                }
                const ncclResult_t rc2 = rccl->GroupEnd();
                bool rccl_bad = false;  // this rank's part of the exchange failed
                auto rccl_fail = [&](const std::string& m) {
                    rccl_bad = true;
                    std::lock_guard<std::mutex> lk(err_mutex);
                    if (rccl_note.empty()) rccl_note = m;
                    rccl_failed.fetch_add(1);
                };
                if (rc != ncclSuccess || rc2 != ncclSuccess) rccl_fail(std::string("RCCL gather: ") + rccl->GetErrorString(rc != ncclSuccess ? rc : rc2));
                if (rank == 0 && !failed && !rccl_bad && copied) {
                    if (rbrt_hip_unpack_tiles_strided(dev, stream, d_slots, img.width, img.height, uint32_t(world), slot_pixels, d_img, d_rgb) != RBRT_OK)
                        fail(rbrt_hip_last_error());
                }
                // Wait for the exchange, but not for ever: if a peer failed inside the group (its send or receive was
                // never enqueued) this rank's side can not complete; it then aborts its communicator instead of hanging.
                for (;;) {
                    const hipError_t q = hipStreamQuery(stream);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) {
                        hip_ok(q, "gather");
                        break;
                    }
                    if (n_failed.load() != 0 || rccl_failed.load() != 0) {
                        if (!rccl_bad) rccl_fail("RCCL gather abandoned: another rank's part of it failed");
                        (void)rccl->CommAbort(comms[rank]);
                        comms[rank] = nullptr;
                        break;
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
                // Did the exchange work for everyone? Again one decision between two barriers. If not, the tiles are still
                // where they were rendered: every rank takes the host path, in this same process.
                barrier.wait();
                const bool fell_back = rccl_failed.load() != 0;
                barrier.wait();
                if (fell_back) {
                    host_gather();
                } else if (rank == 0 && !failed) {
                    hip_ok(hipMemcpy(img.radiance.data(), d_img, n * sizeof(float), hipMemcpyDeviceToHost), "download");
                    hip_ok(hipMemcpy(img.rgb.data(), d_rgb, n, hipMemcpyDeviceToHost), "download");
                }
            }
        } else {
            host_gather();
        }
        t_gather[rank] = seconds_since(t_g);
        const auto t_rel = std::chrono::steady_clock::now();
        if (d_acc) (void)hipFree(d_acc);
        if (d_rad) (void)hipFree(d_rad);
        if (d_rgb) (void)hipFree(d_rgb);
        if (d_img) (void)hipFree(d_img);
        if (rank == 0 && d_slots) (void)hipFree(d_slots);
        if (stream) (void)hipStreamDestroy(stream);
        rbrt_hip_scene_destroy(hs);
        t_release[rank] = seconds_since(t_rel);
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < world; ++r) threads.emplace_back(worker, r);
    worker(0);
    for (auto& t : threads) t.join();
    for (ncclComm_t cm : comms)
        if (cm) (void)rccl->CommDestroy(cm);
    if (rccl_failed.load() != 0) {
        std::fprintf(stderr, "warning: %s; gathered through host memory instead\n", rccl_note.c_str());
        rep.gather = "host (rccl was asked for: " + rccl_note + ")";
    }
    for (int r = 0; r < world; ++r)
        if (!errors[r].empty()) throw Error("GPU " + std::to_string(device_of(r)) + (cfg.oversubscribe ? " (rank " + std::to_string(r) + ")" : "") + ": " + errors[r]);
    if (!cfg.checkpoint_path.empty()) std::remove(cfg.checkpoint_path.c_str());  // (only reached when the render is complete)
    if (!cfg.quiet) std::printf("\rRendering 100%% complete!\n");
    rep.resumed_from_sample = start_sample;
    {   // the slowest rank's set-up, split (the parts of one rank, so that they add up)
        const size_t slow = size_t(std::max_element(t_setup.begin(), t_setup.end()) - t_setup.begin());
        const rbrt_hip_call_times_t& ct = t_create[slow];
        rep.upload_build_s = t_setup[slow];
        rep.hip_init_s = hip_start_s + ct.hip_init_s, rep.upload_s = ct.upload_s, rep.bvh_build_s = ct.bvh_build_s;
        rep.lanes_s = ct.lanes_s + std::max(0.0, ct.create_s - ct.hip_init_s - ct.upload_s - ct.bvh_build_s - ct.lanes_s);
        rep.buffers_s = std::max(0.0, t_setup[slow] - ct.create_s);
        rep.release_s = *std::max_element(t_release.begin(), t_release.end());
    }
    rep.render_s = *std::max_element(t_render.begin(), t_render.end());
    rep.gather_s = *std::max_element(t_gather.begin(), t_gather.end());
    if (cfg.report) *cfg.report = rep;
    return img;
}

}  // namespace rbrt
