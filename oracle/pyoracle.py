"""ctypes loader for the CPU oracle (oracle/rbrt_oracle.cpp). TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
rbrt_amd/. Builds oracle/librbrt_oracle.so with `make -C oracle` when it is missing.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from rbrt_amd import abi  # noqa: E402  (struct definitions only: the ABI header's Python mirror)

LIB = HERE / "librbrt_oracle.so"
f32p, u8p, i32p = abi.f32p, abi.u8p, abi.i32p
_lib = None


def build() -> None:
    subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    src = HERE / "rbrt_oracle.cpp"
    if not LIB.exists() or (src.exists() and LIB.stat().st_mtime < src.stat().st_mtime):
        build()
    L = C.CDLL(str(LIB))
    L.rbrt_oracle_render_window.restype = C.c_uint64
    L.rbrt_oracle_render_window.argtypes = [C.POINTER(abi.Camera), C.POINTER(abi.Scene), C.POINTER(abi.RenderOpts),
                                            C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, f32p, u8p]
    L.rbrt_oracle_render.restype = C.c_uint64
    L.rbrt_oracle_render.argtypes = [C.POINTER(abi.Camera), C.POINTER(abi.Scene), C.POINTER(abi.RenderOpts),
                                     C.c_int, f32p, u8p]
    L.rbrt_oracle_nan_discriminants.restype = C.c_uint64
    L.rbrt_oracle_trace_rays.restype = None
    L.rbrt_oracle_trace_rays.argtypes = [C.POINTER(abi.Scene), f32p, C.c_size_t, C.c_float, C.c_float, C.c_int,
                                         f32p, i32p, i32p, f32p]
    L.rbrt_oracle_camera_new.restype = None
    L.rbrt_oracle_camera_new.argtypes = [f32p, f32p, f32p, C.c_uint32, C.c_uint32, C.c_float, C.POINTER(abi.Camera)]
    L.rbrt_oracle_mesh_prep.restype = C.c_uint32
    L.rbrt_oracle_mesh_prep.argtypes = [f32p, C.c_uint32, C.c_float, f32p, f32p] + [f32p] * 12 + [u8p, f32p, f32p]
    L.rbrt_oracle_kat_vec3.restype = None
    L.rbrt_oracle_kat_vec3.argtypes = [C.c_int, f32p, f32p, f32p]
    L.rbrt_oracle_kat_triangle_normal.restype = None
    L.rbrt_oracle_kat_triangle_normal.argtypes = [f32p, f32p]
    L.rbrt_oracle_kat_refract.restype = C.c_int
    L.rbrt_oracle_kat_refract.argtypes = [f32p, f32p, C.c_float, f32p]
    L.rbrt_oracle_kat_schlick.restype = C.c_float
    L.rbrt_oracle_kat_schlick.argtypes = [C.c_float, C.c_float]
    L.rbrt_oracle_kat_sphere.restype = C.c_int
    L.rbrt_oracle_kat_sphere.argtypes = [C.POINTER(abi.Sphere), f32p, C.c_float, C.c_float, f32p, f32p, f32p]
    L.rbrt_oracle_kat_unit_sphere.restype = None
    L.rbrt_oracle_kat_unit_sphere.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, f32p]
    L.rbrt_oracle_kat_rng.restype = None
    L.rbrt_oracle_kat_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), f32p]
    L.rbrt_oracle_kat_avx.restype = None
    L.rbrt_oracle_kat_avx.argtypes = [C.c_int, f32p, f32p, f32p]
    L.rbrt_oracle_kat_bbox_hit.restype = C.c_int
    L.rbrt_oracle_kat_bbox_hit.argtypes = [f32p, f32p, f32p]
    L.rbrt_oracle_kat_mesh_intersect.restype = C.c_int
    L.rbrt_oracle_kat_mesh_intersect.argtypes = [C.POINTER(abi.Mesh), f32p, C.c_float, f32p, i32p, f32p]
    L.rbrt_oracle_kat_camera_ray.restype = None
    L.rbrt_oracle_kat_camera_ray.argtypes = [C.POINTER(abi.Camera), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, f32p]
    L.rbrt_oracle_kat_scatter.restype = C.c_int
    L.rbrt_oracle_kat_scatter.argtypes = [C.POINTER(abi.Material), f32p, f32p, f32p, C.c_uint64, C.c_uint32,
                                          C.c_uint32, f32p, f32p]
    L.rbrt_oracle_kat_scatter_state.restype = C.c_int
    L.rbrt_oracle_kat_scatter_state.argtypes = [C.POINTER(abi.Material), f32p, f32p, f32p, C.c_uint64, C.c_uint32,
                                                C.c_uint32, f32p, f32p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.rbrt_oracle_kat_basic_triangle.restype = C.c_int
    L.rbrt_oracle_kat_basic_triangle.argtypes = [C.POINTER(abi.Triangle), f32p, C.c_float, C.c_float, f32p, f32p, f32p]
    L.rbrt_oracle_kat_quantise.restype = C.c_uint8
    L.rbrt_oracle_kat_quantise.argtypes = [C.c_float]
    _lib = L
    return L


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(f32p)


# ---- scene preparation ---------------------------------------------------------------------

def camera_new(position, look_at, up, height, width, focal_mm) -> abi.Camera:
    cam = abi.Camera()
    lib().rbrt_oracle_camera_new(_p(_f(position)), _p(_f(look_at)), _p(_f(up)), int(height), int(width),
                                 float(focal_mm), C.byref(cam))
    return cam


def mesh_prep(tri_vertices, scale=1.0, rotation=(0, 0, 0), translation=(0, 0, 0),
              mat: abi.Material | None = None) -> abi.MeshData:
    """tri_vertices: (N,3,3) raw .obj triangles -> MeshData (mesh.rs:41-74,102-112,123-181)."""
    tv = _f(np.asarray(tri_vertices).reshape(-1, 9))
    n = tv.shape[0]
    n_total = n + n % 8
    arrs = {k: np.zeros(n_total, np.float32) for k in abi.MeshData.FIELDS}
    pad = np.zeros(n_total, np.uint8)
    lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
    got = lib().rbrt_oracle_mesh_prep(_p(tv), n, float(scale), _p(_f(rotation)), _p(_f(translation)),
                                      *[_p(arrs[k]) for k in abi.MeshData.FIELDS],
                                      pad.ctypes.data_as(u8p), _p(lo), _p(hi))
    assert got == n_total
    return abi.MeshData(arrs, pad, n, lo, hi, mat if mat is not None else abi.material(abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5)))


# ---- rendering -------------------------------------------------------------------------------

def render(cam: abi.Camera, scene, opts: abi.RenderOpts, n_threads: int = 0, window=None,
           want_rgb8: bool = True, col_stride: int = 1):
    """Returns (radiance[H,W,3] f32, rgb8[H,W,3] u8 or None, n_rays)."""
    H, W = cam.img_height_pix, cam.img_width_pix
    rad = np.zeros((H, W, 3), np.float32)
    rgb = np.zeros((H, W, 3), np.uint8) if want_rgb8 else None
    c0, c1, r0, r1 = window if window is not None else (0, W, 0, H)
    rays = lib().rbrt_oracle_render_window(C.byref(cam), scene.ptr(), C.byref(opts), c0, c1, r0, r1, col_stride, n_threads,
                                           _p(rad), rgb.ctypes.data_as(u8p) if want_rgb8 else None)
    return rad, rgb, int(rays)


def trace_rays(scene: abi.SceneData, rays, min_dist=0.001, max_dist=2000.0, n_threads=0):
    rays = _f(rays).reshape(-1, 6)
    n = rays.shape[0]
    t = np.zeros(n, np.float32)
    dist = np.zeros(n, np.float32)
    obj = np.zeros(n, np.int32)
    tri = np.zeros(n, np.int32)
    lib().rbrt_oracle_trace_rays(scene.ptr(), _p(rays), n, min_dist, max_dist, n_threads, _p(t),
                                 obj.ctypes.data_as(i32p), tri.ctypes.data_as(i32p), _p(dist))
    return t, obj, tri, dist
