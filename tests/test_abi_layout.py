"""The C-ABI library loads without a GPU, exports every symbol include/rbrt_hip.h declares, and the
ctypes mirror (rbrt_amd/abi.py) has the same struct layout a C compiler gives the header."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

from rbrt_amd import abi, tiles

ROOT = Path(__file__).resolve().parent.parent
STRUCTS = {"rbrt_material_t": abi.Material, "rbrt_sphere_t": abi.Sphere, "rbrt_mesh_t": abi.Mesh,
           "rbrt_scene_t": abi.Scene, "rbrt_triangle_t": abi.Triangle, "rbrt_camera_t": abi.Camera, "rbrt_render_opts_t": abi.RenderOpts,
           "rbrt_hip_stats_t": abi.Stats, "rbrt_hip_scene_info_t": abi.SceneInfo}


def _declared(header_name):
    header = (ROOT / "include" / header_name).read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)  # prose in comments may mention calls
    return set(re.findall(r"\b(rbrt_(?:hip_)?[a-z_0-9]+)\s*\(", header)) - {"rbrt_hip_scene"}  # (the opaque struct tag)


def test_library_exports_every_declared_symbol():
    """Both headers: the drop-in boundary (rbrt_hip.h) and the test hooks / diagnostics (rbrt_hip_debug.h)."""
    lib = abi.load_hip()
    for header_name, table in (("rbrt_hip.h", abi.HIP_SYMBOLS), ("rbrt_hip_debug.h", abi.DEBUG_SYMBOLS)):
        declared = _declared(header_name)
        for name in sorted(declared):
            assert hasattr(lib, name), f"{name} is declared in {header_name} but not exported"
        assert declared == set(table), (header_name, declared ^ set(table))
    assert lib.rbrt_hip_abi_version() == 2


def test_drop_in_header_has_no_test_hooks():
    """What a Rust host binds is the render path only: hooks and diagnostics live in rbrt_hip_debug.h."""
    declared = _declared("rbrt_hip.h")
    assert not [n for n in declared if "selftest" in n or "debug" in n or "trace_rays" in n or "bvh_build" in n or "timing" in n]


def test_struct_layout_matches_the_c_header(tmp_path):
    fields = []
    for cname, cls in STRUCTS.items():
        fields.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _ in cls._fields_:
            fields.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "rbrt_hip.h"\nint main(void){' + "".join(fields) + "return 0;}"
    (tmp_path / "layout.c").write_text(src)
    subprocess.run(["gcc", "-I", str(ROOT / "include"), "-o", str(tmp_path / "layout"), str(tmp_path / "layout.c")],
                   check=True)
    out = subprocess.run([str(tmp_path / "layout")], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for cname, cls in STRUCTS.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, f"{cname}.{f}"


def test_defaults_match_the_reference_constants():
    lib = abi.load_hip()
    o = abi.RenderOpts()
    lib.rbrt_render_opts_default(C.byref(o))
    assert (o.spp, o.max_depth) == (5, 50)          # src/main.rs:47, lib.rs:99
    assert (o.min_dist, o.max_dist) == (C.c_float(0.001).value, 2000.0)   # lib.rs:44-45
    assert list(o.bg) == [C.c_float(0.05).value, C.c_float(0.05).value, C.c_float(0.8).value]  # lib.rs:89-93
    assert (o.tile_rank, o.tile_world, o.flags) == (0, 1, 0)


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU every render entry point fails loudly (this container has none)."""
    import rbrt_amd
    if rbrt_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    cam = abi.Camera()
    cam.img_width_pix, cam.img_height_pix = 8, 8
    with pytest.raises(abi.RbrtError) as e:
        rbrt_amd.render_scene(cam, 1, abi.SceneData())
    assert e.value.code == abi.RBRT_ERR_NO_DEVICE
    with pytest.raises(abi.RbrtError):
        rbrt_amd.HipScene(abi.SceneData())


def test_invalid_arguments_are_rejected_before_touching_the_device():
    lib = abi.load_hip()
    h = C.c_void_p()
    assert lib.rbrt_hip_scene_create(None, 0, C.byref(h)) == abi.RBRT_ERR_INVALID_ARG
    sc = abi.SceneData(spheres=[((0, 0, -5), 1.0, abi.Material(7, (C.c_float * 3)(0, 0, 0), 0.0))])
    assert lib.rbrt_hip_scene_create(sc.ptr(), 0, C.byref(h)) == abi.RBRT_ERR_INVALID_ARG  # unknown material kind
    assert b"material" in lib.rbrt_hip_last_error()


@pytest.mark.parametrize("w,h", [(1, 1), (8, 8), (9, 17), (100, 60), (1024, 768), (1920, 1080)])
def test_packed_pixels_matches_python_index_math(w, h):
    lib = abi.load_hip()
    for world in (1, 2, 3, 8):
        total = 0
        for r in range(world):
            assert lib.rbrt_hip_packed_pixels(w, h, r, world) == tiles.packed_pixels(w, h, r, world)
            total += tiles.packed_pixels(w, h, r, world)
        assert total == tiles.n_tiles(w, h) * 64


def test_argument_errors_come_back_as_codes_without_a_gpu():
    """Error behaviour at the boundary: negative status codes + rbrt_hip_last_error(), never an abort. These
    checks run before any device is touched, so they hold on a machine without a GPU too."""
    lib = abi.load_hip()
    out = C.c_void_p()
    assert lib.rbrt_hip_scene_create(None, 0, C.byref(out)) == -1  # RBRT_ERR_INVALID_ARG
    assert b"null" in lib.rbrt_hip_last_error()
    lam = abi.material(abi.MAT_LAMBERTIAN, (0.5, 0.5, 0.5))
    too_many = abi.SceneData(spheres=[((float(i), 0.0, 0.0), 0.25, lam) for i in range(256)])
    assert lib.rbrt_hip_scene_create(too_many.ptr(), 0, C.byref(out)) == -5  # RBRT_ERR_UNSUPPORTED: > 255 objects
    assert b"255" in lib.rbrt_hip_last_error()
    bad_mat = abi.SceneData(spheres=[((0.0, 0.0, 0.0), 1.0, abi.Material(7, abi._f3((0, 0, 0)), 0.0))])
    assert lib.rbrt_hip_scene_create(bad_mat.ptr(), 0, C.byref(out)) == -1
    opts = abi.default_opts(spp=1)
    assert lib.rbrt_hip_render_device(None, None, C.byref(opts), None, None, None) == -1
    assert lib.rbrt_hip_scene_set_pipeline(None, 2) == -1
    assert lib.rbrt_hip_scene_destroy(None) == 0  # destroying nothing is fine


def test_the_dealing_of_tiles_is_one_bijection_in_the_library_and_in_python():
    """rbrt_hip.h "How tiles are dealt to ranks": tile number <-> (tile row, tile column), rows rotated by RBRT_TILE_SKEW per
    row. The library's two directions are inverse to each other for every tiles_x (smaller than, equal to, not coprime with
    the skew), cover every tile exactly once, and agree with rbrt_amd/tiles.py, which the CPU multi-rank tests pack with."""
    import ctypes as C

    import numpy as np
    from rbrt_amd import tiles
    lib = abi.load_hip()
    for tiles_x, tiles_y in ((1, 5), (2, 7), (3, 4), (13, 8), (128, 96), (240, 135)):
        seen = set()
        for t in range(tiles_x * tiles_y):
            ty, tx = C.c_uint32(), C.c_uint32()
            lib.rbrt_hip_tile_xy(t, tiles_x, C.byref(ty), C.byref(tx))
            assert ty.value == t // tiles_x and tx.value == (t % tiles_x + tiles.SKEW * ty.value) % tiles_x
            assert lib.rbrt_hip_tile_number(ty.value, tx.value, tiles_x) == t
            seen.add((ty.value, tx.value))
        assert len(seen) == tiles_x * tiles_y
    w, h, world = 100, 60, 3
    owner, slot = tiles._index_maps(w, h, world)
    tx_n = (w + 7) // 8
    for (y, x) in ((0, 0), (59, 99), (17, 42), (8, 8), (40, 96)):
        t = lib.rbrt_hip_tile_number(y // 8, x // 8, tx_n)
        assert owner[y, x] == t % world and slot[y, x] == (t // world) * 64 + (y % 8) * 8 + x % 8
    # with 8 ranks and a width that is a multiple of 8 tiles, a rank's tiles are no longer whole columns
    owner, _ = tiles._index_maps(1024, 768, 8)
    assert all(len(np.unique(owner[::8, c * 8])) == 8 for c in (0, 5, 127))
