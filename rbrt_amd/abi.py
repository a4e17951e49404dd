"""ctypes mirror of include/rbrt_hip.h (the C ABI of the HIP hot path).

Python here is plumbing for tests and bench.py: it declares the same POD structs a Rust
`extern "C"` block or the C++ host would, loads `rbrt_amd/lib/librbrt_hip.so`, and checks return
codes. There is no Python or CPU implementation of the render path behind it: if the shared
library is missing, `load_hip()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB_HIP = ROOT / "rbrt_amd" / "lib" / "librbrt_hip.so"
LIB_HOST = ROOT / "rbrt_amd" / "lib" / "librbrt_host.so"

RBRT_OK = 0
RBRT_ERR_INVALID_ARG = -1
RBRT_ERR_NO_DEVICE = -2
RBRT_ERR_HIP = -3
RBRT_ERR_OOM = -4
RBRT_ERR_UNSUPPORTED = -5
RBRT_ERR_NAN = -6

MAT_LAMBERTIAN = 0
MAT_METAL = 1
MAT_DIELECTRIC = 2

FLAG_COLLECT_STATS = 1
TILE = 8

f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo", C.c_float * 3), ("param", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("mat", Material)]


class Mesh(C.Structure):
    _fields_ = [
        ("n_total", C.c_uint32),
        ("n_real", C.c_uint32),
        ("v0x", f32p), ("v0y", f32p), ("v0z", f32p),
        ("e1x", f32p), ("e1y", f32p), ("e1z", f32p),
        ("e2x", f32p), ("e2y", f32p), ("e2z", f32p),
        ("nx", f32p), ("ny", f32p), ("nz", f32p),
        ("is_padding", u8p),
        ("bbox_lo", C.c_float * 3),
        ("bbox_hi", C.c_float * 3),
        ("mat", Material),
    ]


class Triangle(C.Structure):
    _fields_ = [("corners", (C.c_float * 3) * 3), ("mat", Material)]


class Scene(C.Structure):
    _fields_ = [
        ("n_spheres", C.c_uint32),
        ("spheres", C.POINTER(Sphere)),
        ("n_meshes", C.c_uint32),
        ("meshes", C.POINTER(Mesh)),
        ("n_triangles", C.c_uint32),
        ("triangles", C.POINTER(Triangle)),
        ("element_order", C.POINTER(C.c_uint32)),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("right", C.c_float * 3),
        ("up", C.c_float * 3),
        ("img_center_point", C.c_float * 3),
        ("mm_per_pix_hor", C.c_float),
        ("mm_per_pix_vert", C.c_float),
        ("img_width_pix", C.c_uint32),
        ("img_height_pix", C.c_uint32),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("spp", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("min_dist", C.c_float),
        ("max_dist", C.c_float),
        ("bg", C.c_float * 3),
        ("seed", C.c_uint64),
        ("tile_rank", C.c_uint32),
        ("tile_world", C.c_uint32),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("mesh_gate_pass", C.c_uint64),
        ("nodes_visited", C.c_uint64),
        ("tris_tested", C.c_uint64),
        ("mesh_hits", C.c_uint64),
        ("samples", C.c_uint64),
        ("nan_discriminants", C.c_uint64),
        ("node_bytes", C.c_uint32),
        ("tri_bytes", C.c_uint32),
    ]


class SceneInfo(C.Structure):
    _fields_ = [
        ("n_spheres", C.c_uint32), ("n_meshes", C.c_uint32),
        ("n_meshes_device_built", C.c_uint32), ("bvh_stack_need", C.c_uint32),
        ("n_nodes", C.c_uint64), ("n_triangles", C.c_uint64),
        ("trace_waves", C.c_uint32), ("lds_bytes_per_wave", C.c_uint32),
        ("occupancy_api_waves_per_cu", C.c_uint32), ("n_cus", C.c_uint32),
    ]


class CallTimes(C.Structure):  # rbrt_hip_call_times_t (include/rbrt_hip_debug.h)
    _fields_ = [(k, C.c_double) for k in ("hip_init_s", "upload_s", "bvh_build_s", "lanes_s", "create_s", "render_s", "copy_s",
                                          "destroy_s", "total_s")] + [("meshes_device_built", C.c_uint32), ("meshes_host_built", C.c_uint32)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/rbrt_hip.h (the drop-in boundary) declares: name -> (restype, argtypes)
HIP_SYMBOLS = {
    "rbrt_hip_render": (C.c_int, [C.POINTER(Camera), C.POINTER(Scene), C.POINTER(RenderOpts), f32p, u8p]),
    "rbrt_hip_scene_create": (C.c_int, [C.POINTER(Scene), C.c_int, C.POINTER(C.c_void_p)]),
    "rbrt_hip_scene_destroy": (C.c_int, [C.c_void_p]),
    "rbrt_hip_packed_pixels": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rbrt_hip_tile_xy": (None, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rbrt_hip_tile_number": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "rbrt_hip_render_device": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "rbrt_hip_render_pass": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderOpts), C.c_void_p, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "rbrt_hip_unpack_tiles": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p]),
    "rbrt_hip_unpack_tiles_strided": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_size_t, C.c_void_p, C.c_void_p]),
    "rbrt_hip_scene_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "rbrt_hip_scene_check": (C.c_int, [C.c_void_p]),
    "rbrt_render_opts_default": (None, [C.POINTER(RenderOpts)]),
    "rbrt_hip_device_count": (C.c_int, []),
    "rbrt_hip_last_error": (C.c_char_p, []),
    "rbrt_hip_abi_version": (C.c_int, []),
    "rbrt_hip_scene_set_pipeline": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rbrt_hip_scene_info": (C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
}
# ... and include/rbrt_hip_debug.h (test hooks and diagnostics, same library)
DEBUG_SYMBOLS = {
    "rbrt_hip_scene_last_batching": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rbrt_hip_trace_rays": (C.c_int, [C.c_void_p, f32p, C.c_size_t, C.c_float, C.c_float, f32p, i32p, i32p, f32p]),
    "rbrt_hip_selftest_ieee": (C.c_int, [C.c_uint64, C.c_size_t, C.POINTER(C.c_uint64)]),
    "rbrt_hip_selftest_gate": (C.c_int, [f32p, f32p, f32p, C.c_size_t, u8p, u8p]),
    "rbrt_hip_bvh_build_host": (C.c_int, [C.POINTER(Mesh), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32),
                                          f32p]),
    "rbrt_hip_free_host": (None, [C.c_void_p]),
    "rbrt_hip_bvh_build_host_records": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                                  C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), f32p]),
    "rbrt_hip_bvh_build_device": (C.c_int, [C.POINTER(Mesh), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32),
                                            f32p, C.POINTER(C.c_int)]),
    "rbrt_hip_scene_debug_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]),
    "rbrt_hip_scene_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "rbrt_hip_scene_kernel_ms": (C.c_int, [C.c_void_p, f32p, f32p, C.POINTER(C.c_uint32)]),
    "rbrt_hip_scene_debug_set_counter": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64]),
    "rbrt_hip_scene_launch_mix": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rbrt_hip_debug_scatter": (C.c_int, [C.POINTER(Material), f32p, f32p, f32p, C.POINTER(C.c_uint32), C.c_size_t, f32p, u8p,
                                        C.POINTER(C.c_uint32)]),
    "rbrt_hip_debug_primary_cull": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(C.c_uint32), C.c_size_t]),
    "rbrt_hip_scene_helper_launches": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "rbrt_hip_scene_create_times": (C.c_int, [C.c_void_p, C.POINTER(CallTimes)]),
    "rbrt_hip_last_render_times": (C.c_int, [C.POINTER(CallTimes)]),
    "rbrt_hip_scene_refine_wait": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
}


class RbrtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rbrt_hip error {code}: {msg}")
        self.code = code


_hip = None


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.

    The PyTorch wheel bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and finds it by file
    name, so a process that loaded /opt/rocm's copy first ends up with two HIP/HSA runtimes and
    torch then reports "No HIP GPUs are available". bench.py and some tests use torch for device
    buffers and torch.distributed next to this library, so when torch is installed its runtime is
    loaded first and librbrt_hip.so's NEEDED libamdhip64.so.7 binds to that same copy. Without
    torch (the C++ host, a Rust host) the system runtime under /opt/rocm is used as usual.
    """
    if os.environ.get("RBRT_NO_TORCH_PRELOAD"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
        if cand.exists():
            C.CDLL(str(cand), mode=getattr(os, "RTLD_GLOBAL", 0x100) | getattr(os, "RTLD_NOW", 2))
    except OSError:
        pass


def load_hip() -> C.CDLL:
    """Load the product library. Fails loudly when it has not been built: no fallback exists."""
    global _hip
    if _hip is not None:
        return _hip
    lib_path = Path(os.environ.get("RBRT_HIP_LIB", LIB_HIP))  # override: A/B builds of the same ABI
    if not lib_path.exists():
        raise FileNotFoundError(
            f"{lib_path} is missing: build it with `make` (or __graft_entry__.build()). "
            "rbrt_amd has no CPU or pure-Python render path.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(str(lib_path), mode=getattr(os, "RTLD_NOW", 2))
    for name, (res, args) in {**HIP_SYMBOLS, **DEBUG_SYMBOLS}.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _hip = lib
    return lib


def check(code: int) -> None:
    if code != RBRT_OK:
        raise RbrtError(code, load_hip().rbrt_hip_last_error().decode(errors="replace"))


# ------------------------------------------------------------------------------------------
# helpers to build the POD structs from Python values (keeping the backing arrays alive)
# ------------------------------------------------------------------------------------------

def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def material(kind: int, albedo=(0.0, 0.0, 0.0), param: float = 0.0) -> Material:
    return Material(kind, _f3(albedo), float(param))


def fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(f32p)


class SceneData:
    """Owns the numpy arrays and ctypes arrays a `Scene` struct points into."""

    def __init__(self, spheres=(), meshes=(), triangles=(), element_order=None):
        # spheres: iterable of (center, radius, Material); meshes: iterable of MeshData; triangles: iterable of
        # (corners (3,3), Material); element_order: None or entries as in rbrt_scene_t.element_order
        self.spheres = list(spheres)
        self.meshes = list(meshes)
        self.triangles = list(triangles)
        self.element_order = None if element_order is None else [int(x) for x in element_order]
        self._sph = (Sphere * max(1, len(self.spheres)))()
        for i, (c, r, m) in enumerate(self.spheres):
            self._sph[i] = Sphere(_f3(c), float(r), m)
        self._msh = (Mesh * max(1, len(self.meshes)))()
        for i, md in enumerate(self.meshes):
            self._msh[i] = md.struct
        self._tri = (Triangle * max(1, len(self.triangles)))()
        for i, (corners, m) in enumerate(self.triangles):
            for a in range(3):
                for b in range(3):
                    self._tri[i].corners[a][b] = float(corners[a][b])
            self._tri[i].mat = m
        self._order = None if self.element_order is None else (C.c_uint32 * len(self.element_order))(*self.element_order)
        self.struct = Scene(len(self.spheres), self._sph, len(self.meshes), self._msh, len(self.triangles), self._tri,
                            self._order if self._order is not None else C.POINTER(C.c_uint32)())

    def ptr(self):
        return C.byref(self.struct)


class MeshData:
    """SoA arrays of one mesh in the reference's layout (mesh.rs:12-25)."""

    FIELDS = ("v0x", "v0y", "v0z", "e1x", "e1y", "e1z", "e2x", "e2y", "e2z", "nx", "ny", "nz")

    def __init__(self, arrays: dict, is_padding: np.ndarray, n_real: int, bbox_lo, bbox_hi, mat: Material):
        self.arrays = {k: np.ascontiguousarray(arrays[k], dtype=np.float32) for k in self.FIELDS}
        self.is_padding = np.ascontiguousarray(is_padding, dtype=np.uint8)
        n_total = len(self.is_padding)
        for k in self.FIELDS:
            assert len(self.arrays[k]) == n_total, k
        self.n_total, self.n_real = n_total, int(n_real)
        self.bbox_lo, self.bbox_hi = np.float32(bbox_lo), np.float32(bbox_hi)
        self.struct = Mesh()
        self.struct.n_total = n_total
        self.struct.n_real = self.n_real
        for k in self.FIELDS:
            setattr(self.struct, k, fptr(self.arrays[k]))
        self.struct.is_padding = self.is_padding.ctypes.data_as(u8p)
        self.struct.bbox_lo = _f3(bbox_lo)
        self.struct.bbox_hi = _f3(bbox_hi)
        self.struct.mat = mat


def default_opts(spp: int = 5, seed: int = 1, **kw) -> RenderOpts:
    o = RenderOpts()
    o.spp, o.max_depth, o.min_dist, o.max_dist = spp, 50, 0.001, 2000.0
    o.bg = _f3((0.05, 0.05, 0.8))
    o.seed, o.tile_rank, o.tile_world, o.flags, o.reserved = seed, 0, 1, 0, 0
    for k, v in kw.items():
        if k == "bg":
            o.bg = _f3(v)
        else:
            setattr(o, k, v)
    return o


# ------------------------------------------------------------------------------------------
# C++ host library (rbrt_amd/host): YAML scene -> camera + SoA meshes, PNG writer
# ------------------------------------------------------------------------------------------
_host = None


def load_host() -> C.CDLL:
    global _host
    if _host is not None:
        return _host
    if not LIB_HOST.exists():
        raise FileNotFoundError(f"{LIB_HOST} is missing: build it with `make`")
    lib = C.CDLL(str(LIB_HOST))
    lib.rbrt_host_last_error.restype = C.c_char_p
    lib.rbrt_host_scene_load.restype = C.c_int
    lib.rbrt_host_scene_load.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.rbrt_host_scene_camera.restype = C.POINTER(Camera)
    lib.rbrt_host_scene_camera.argtypes = [C.c_void_p]
    lib.rbrt_host_scene_scene.restype = C.POINTER(Scene)
    lib.rbrt_host_scene_scene.argtypes = [C.c_void_p]
    lib.rbrt_host_scene_free.restype = None
    lib.rbrt_host_scene_free.argtypes = [C.c_void_p]
    lib.rbrt_host_save_image.restype = C.c_int
    lib.rbrt_host_save_image.argtypes = [C.c_char_p, u8p, C.c_uint32, C.c_uint32]
    lib.rbrt_host_write_png.restype = C.c_int
    lib.rbrt_host_write_png.argtypes = [C.c_char_p, u8p, C.c_uint32, C.c_uint32]
    lib.rbrt_host_camera_new.restype = None
    lib.rbrt_host_camera_new.argtypes = [f32p, f32p, f32p, C.c_uint32, C.c_uint32, C.c_float, C.POINTER(Camera)]
    _host = lib
    return lib


class HostScene:
    """A scene loaded by the C++ host from a YAML file (src/main.rs:70-80). Quacks like SceneData:
    `.ptr()` gives the rbrt_scene_t*, `.camera` the rbrt_camera_t for the requested image size."""

    def __init__(self, yaml_path, height: int, width: int):
        self._lib = load_host()
        self._h = C.c_void_p()
        rc = self._lib.rbrt_host_scene_load(str(yaml_path).encode(), height, width, C.byref(self._h))
        if rc != 0:
            raise RuntimeError("rbrt_host_scene_load: " + self._lib.rbrt_host_last_error().decode(errors="replace"))
        self.camera = Camera()
        C.memmove(C.byref(self.camera), self._lib.rbrt_host_scene_camera(self._h), C.sizeof(Camera))
        self.struct = self._lib.rbrt_host_scene_scene(self._h).contents

    def ptr(self):
        return C.byref(self.struct)

    def mesh_arrays(self, i: int) -> dict:
        m = self.struct.meshes[i]
        out = {k: np.ctypeslib.as_array(getattr(m, k), (m.n_total,)).copy() for k in MeshData.FIELDS}
        out["is_padding"] = np.ctypeslib.as_array(m.is_padding, (m.n_total,)).copy()
        out["bbox_lo"] = np.array(list(m.bbox_lo), np.float32)
        out["bbox_hi"] = np.array(list(m.bbox_hi), np.float32)
        out["n_real"] = m.n_real
        return out

    def close(self):
        if self._h:
            self._lib.rbrt_host_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def save_image(path, rgb8: np.ndarray) -> None:
    """rbrt::ImageBuffer::save: encoder by file extension (src/main.rs:86)."""
    h, w, _ = rgb8.shape
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    lib = load_host()
    if lib.rbrt_host_save_image(str(path).encode(), rgb8.ctypes.data_as(u8p), w, h) != 0:
        raise RuntimeError(lib.rbrt_host_last_error().decode(errors="replace"))


def write_png(path, rgb8: np.ndarray) -> None:
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    h, w, _ = rgb8.shape
    if load_host().rbrt_host_write_png(str(path).encode(), rgb8.ctypes.data_as(u8p), w, h) != 0:
        raise RuntimeError(load_host().rbrt_host_last_error().decode(errors="replace"))
