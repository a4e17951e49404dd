"""rbrt_amd — MI355X (gfx950) implementation of baurst/rbrt's path-tracing hot path.

The product is the C ABI in include/rbrt_hip.h (rbrt_amd/lib/librbrt_hip.so: hand-written HIP
kernels + host BVH builder) and the C++ host in rbrt_amd/host (CLI, YAML, .obj, PNG). This Python
package is a thin ctypes binding over those for tests and bench.py.

    render_scene(cam, num_samples, scene)  <->  rbrt_lib::render_scene (lib.rs:75-79)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi
from .abi import (Camera, Material, RenderOpts, RbrtError, SceneData, MeshData, default_opts, load_hip,  # noqa: F401
                  material, MAT_DIELECTRIC, MAT_LAMBERTIAN, MAT_METAL)

__all__ = ["render_scene", "HipScene", "abi", "device_count"]


def device_count() -> int:
    return int(load_hip().rbrt_hip_device_count())


def render_scene(cam: abi.Camera, num_samples: int, scene: abi.SceneData, seed: int = 1, want_radiance: bool = True,
                 **opt_overrides):
    """One-shot render through rbrt_hip_render (host buffers in and out).

    Same contract as the reference's render_scene (lib.rs:75-79) plus the pre-gamma radiance:
    returns (radiance float32[H,W,3], rgb8 uint8[H,W,3]); want_radiance=False: (None, rgb8), exactly what the
    reference returns.
    """
    lib = load_hip()
    opts = default_opts(spp=num_samples, seed=seed, **opt_overrides)
    H, W = cam.img_height_pix, cam.img_width_pix
    rad = np.zeros((H, W, 3), np.float32) if want_radiance else None
    rgb = np.empty((H, W, 3), np.uint8) if not want_radiance else np.zeros((H, W, 3), np.uint8)
    rc = lib.rbrt_hip_render(C.byref(cam), scene.ptr(), C.byref(opts), rad.ctypes.data_as(abi.f32p) if want_radiance else None,
                             rgb.ctypes.data_as(abi.u8p))
    abi.check(rc)
    return rad, rgb


def last_render_times() -> dict:
    """Where the time of this thread's last one-shot render_scene (rbrt_hip_render) went (rbrt_hip_last_render_times)."""
    t = abi.CallTimes()
    abi.check(load_hip().rbrt_hip_last_render_times(C.byref(t)))
    return t.as_dict()


class HipScene:
    """Device-resident scene (rbrt_hip_scene_create): upload + BVH build once, render many times."""

    def __init__(self, scene: abi.SceneData, device: int = 0):
        self._lib = load_hip()
        self._h = C.c_void_p()
        self.device = device
        self.scene = scene  # keep host arrays alive
        abi.check(self._lib.rbrt_hip_scene_create(scene.ptr(), device, C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.rbrt_hip_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def render_device(self, cam: abi.Camera, opts: abi.RenderOpts, d_radiance: int | None, d_rgb8: int | None = None,
                      stream: int | None = None):
        """Asynchronous render into device pointers (ints, e.g. torch.Tensor.data_ptr())."""
        abi.check(self._lib.rbrt_hip_render_device(self._h, C.byref(cam), C.byref(opts), C.c_void_p(stream or 0),
                                                   C.c_void_p(d_radiance or 0), C.c_void_p(d_rgb8 or 0)))

    def render_pass(self, cam: abi.Camera, opts: abi.RenderOpts, sample_begin: int, sample_end: int, d_accum: int,
                    d_radiance: int | None = None, d_rgb8: int | None = None, stream: int | None = None):
        """Samples [sample_begin, sample_end) of opts.spp, accumulated into d_accum (rbrt_hip_render_pass)."""
        abi.check(self._lib.rbrt_hip_render_pass(self._h, C.byref(cam), C.byref(opts), C.c_void_p(stream or 0), sample_begin,
                                                 sample_end, C.c_void_p(d_accum), C.c_void_p(d_radiance or 0),
                                                 C.c_void_p(d_rgb8 or 0)))

    def set_pipeline(self, depth: int):
        """Overlap consecutive trace launches over `depth` internal streams (rbrt_hip_scene_set_pipeline)."""
        abi.check(self._lib.rbrt_hip_scene_set_pipeline(self._h, int(depth)))

    def set_timing(self, on: bool = True):
        abi.check(self._lib.rbrt_hip_scene_set_timing(self._h, int(on)))

    def kernel_ms(self):
        """(total trace-kernel ms, total resolve-kernel ms, number of trace launches) since set_timing(True)."""
        t, r, n = C.c_float(), C.c_float(), C.c_uint32()
        abi.check(self._lib.rbrt_hip_scene_kernel_ms(self._h, C.byref(t), C.byref(r), C.byref(n)))
        return t.value, r.value, n.value

    def launch_mix(self):
        """(full-grid, half-grid) trace launches since set_timing(True) (rbrt_hip_scene_launch_mix)."""
        a, b = C.c_uint32(), C.c_uint32()
        abi.check(self._lib.rbrt_hip_scene_launch_mix(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def helper_launches(self) -> int:
        """Helper launches since set_timing(True) (rbrt_hip_scene_helper_launches)."""
        n = C.c_uint32()
        abi.check(self._lib.rbrt_hip_scene_helper_launches(self._h, C.byref(n)))
        return n.value

    def create_times(self) -> dict:
        """Where the time of rbrt_hip_scene_create went (rbrt_hip_scene_create_times)."""
        t = abi.CallTimes()
        abi.check(self._lib.rbrt_hip_scene_create_times(self._h, C.byref(t)))
        return t.as_dict()

    def refine_wait(self, timeout_s: float = 60.0):
        """Waits for the background build of the host builder's trees and adopts them (rbrt_hip_scene_refine_wait).
        Returns (state, seconds): 0 none was started, 1 in use now, 2 failed / cancelled, 3 still at work."""
        st, sec = C.c_int(), C.c_double()
        abi.check(self._lib.rbrt_hip_scene_refine_wait(self._h, float(timeout_s), C.byref(st), C.byref(sec)))
        return st.value, sec.value

    def info(self) -> dict:
        i = abi.SceneInfo()
        abi.check(self._lib.rbrt_hip_scene_info(self._h, C.byref(i)))
        return {k: int(getattr(i, k)) for k, _ in abi.SceneInfo._fields_}

    def check(self):
        """Synchronise and raise RbrtError if a kernel flagged a NaN sphere discriminant (sphere.rs:33) or corrupt
        path state since the last check (rbrt_hip_scene_check)."""
        abi.check(self._lib.rbrt_hip_scene_check(self._h))

    def last_batches(self):
        """(samples per batch, number of batches) of the last render_device call."""
        b, n = C.c_uint32(), C.c_uint32()
        abi.check(self._lib.rbrt_hip_scene_last_batching(self._h, C.byref(b), C.byref(n)))
        return b.value, n.value

    def stats(self) -> dict:
        st = abi.Stats()
        abi.check(self._lib.rbrt_hip_scene_stats(self._h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in abi.Stats._fields_}

    def set_debug_counter(self, index: int, value: int):
        abi.check(self._lib.rbrt_hip_scene_debug_set_counter(self._h, int(index), int(value)))

    def raw_debug_counters(self) -> list:
        buf = (C.c_uint64 * 64)()
        abi.check(self._lib.rbrt_hip_scene_debug_counters(self._h, buf, 64))
        return [int(x) for x in buf]

    def debug_counters(self) -> dict:
        """Pass statistics of the megakernel from the last counting render (diagnostic)."""
        buf = (C.c_uint64 * 64)()
        abi.check(self._lib.rbrt_hip_scene_debug_counters(self._h, buf, 64))
        names = ("empty", "trav", "term", "lamb", "metal", "diel")
        d = {f"passes_{n}": int(buf[i]) for i, n in enumerate(names)}
        d.update({f"slots_{n}": int(buf[6 + i]) for i, n in enumerate(names)})
        d.update(trav_wave_steps=int(buf[12]), trav_lane_steps=int(buf[13]), refill_rounds=int(buf[14]),
                 sched_rounds=int(buf[15]), cycles_trav=int(buf[16]), cycles_shade=int(buf[17]),
                 cycles_total=int(buf[18]), leaf_rounds=int(buf[19]), leaf_lanes=int(buf[20]),
                 walk_rounds=int(buf[21]), walk_lanes=int(buf[22]), shading_pass_free_lanes=int(buf[23]),
                 rt_first_start=int(buf[24]), rt_last_workout=int(buf[25]), rt_last_end=int(buf[26]),
                 rt_sum_wave_time=int(buf[27]), rt_first_workout=int(buf[28]),
                 path_len_hist=[int(buf[32 + i]) for i in range(8)],     # bounces 0, 1, 2-3, 4-7, ... 64+
                 long_path_objects=[int(buf[40 + i]) for i in range(8)], # bounces of paths >= 16, per object id
                 long_path_total=int(buf[48]), long_path_dielectric=int(buf[47]), shade_extra_rounds=int(buf[29]),
                 drain_slowest=dict(us=int(buf[50]) >> 44, rounds=(int(buf[50]) >> 32) & 0xFFF,
                                    trav_steps=(int(buf[50]) >> 16) & 0xFFFF, passes=int(buf[50]) & 0xFFFF),
                 drain_sum=dict(rounds=int(buf[51]), trav_steps=int(buf[52]), passes=int(buf[53]),
                                lane_steps=int(buf[54])),
                 stack_pushes_beyond_lds=int(buf[30]), stack_deepest=int(buf[31]),
                 parked_at_burst_entry=int(buf[55]), fullest_shading_kind_at_burst_entry=int(buf[56]),
                 shared_entries_given=int(buf[59]), share_rounds=int(buf[60]), traversals_ending_at_root=int(buf[61]),
                 sphere_tail_runs=int(buf[62]), sphere_tail_lanes=int(buf[63]),
                 # what a pass of mixed kinds would pick up: passes (and lanes) where the fullest OTHER scatter kind has >= 8 / >= 16 waiting
                 mixed_pass_8=dict(passes=int(buf[49]) >> 32, lanes=int(buf[49]) & 0xFFFFFFFF),
                 mixed_pass_16=dict(passes=int(buf[58]) >> 32, lanes=int(buf[58]) & 0xFFFFFFFF))
        return d

    def primary_cull(self, cam):
        """The primary-ray culling table for `cam` (rbrt_hip_debug_primary_cull; test hook): uint32 (tiles_y, tiles_x)."""
        tx, ty = (cam.img_width_pix + 7) // 8, (cam.img_height_pix + 7) // 8
        out = np.zeros(tx * ty, np.uint32)
        abi.check(self._lib.rbrt_hip_debug_primary_cull(self._h, C.byref(cam), out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size))
        return out.reshape(ty, tx)

    def trace_rays(self, rays, min_dist=0.001, max_dist=2000.0):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        t = np.zeros(n, np.float32)
        dist = np.zeros(n, np.float32)
        obj = np.zeros(n, np.int32)
        tri = np.zeros(n, np.int32)
        abi.check(self._lib.rbrt_hip_trace_rays(self._h, rays.ctypes.data_as(abi.f32p), n, min_dist, max_dist,
                                                t.ctypes.data_as(abi.f32p), obj.ctypes.data_as(abi.i32p),
                                                tri.ctypes.data_as(abi.i32p), dist.ctypes.data_as(abi.f32p)))
        return t, obj, tri, dist


def debug_scatter(kind, albedo, param, in_dir, point, normal, rng_state):
    """n scatter events through the device functions the shading passes use (rbrt_hip_debug_scatter; test hook).
    Arrays of length n: kind int32, albedo (n,3), param, in_dir (n,3), point (n,3), normal (n,3), rng_state (n,2) uint32.
    Returns (out_dir (n,3) float32, ok (n,) uint8, state_after (n,2) uint32)."""
    n = len(kind)
    mats = (abi.Material * n)()
    for i in range(n):
        mats[i] = abi.material(int(kind[i]), tuple(float(x) for x in albedo[i]), float(param[i]))
    f = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    in_dir, point, normal = f(in_dir), f(point), f(normal)
    st = np.ascontiguousarray(rng_state, np.uint32)
    out_dir, ok, st_out = np.zeros((n, 3), np.float32), np.zeros(n, np.uint8), np.zeros((n, 2), np.uint32)
    u32p = C.POINTER(C.c_uint32)
    abi.check(load_hip().rbrt_hip_debug_scatter(mats, abi.fptr(in_dir), abi.fptr(point), abi.fptr(normal), st.ctypes.data_as(u32p), n,
                                                 abi.fptr(out_dir), ok.ctypes.data_as(abi.u8p), st_out.ctypes.data_as(u32p)))
    return out_dir, ok, st_out


def packed_pixels(width: int, height: int, rank: int, world: int) -> int:
    return int(load_hip().rbrt_hip_packed_pixels(width, height, rank, world))


def unpack_tiles(device: int, d_gathered: int, width: int, height: int, world: int, d_radiance: int | None,
                 d_rgb8: int | None = None, stream: int | None = None, rank_stride_pixels: int = 0):
    """De-interleave gathered per-rank tiles; rank_stride_pixels > 0: equal-size slot per rank."""
    abi.check(load_hip().rbrt_hip_unpack_tiles_strided(device, C.c_void_p(stream or 0), C.c_void_p(d_gathered), width,
                                                       height, world, rank_stride_pixels, C.c_void_p(d_radiance or 0),
                                                       C.c_void_p(d_rgb8 or 0)))
