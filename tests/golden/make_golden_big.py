#!/usr/bin/env python3
"""Golden fixtures for the FULL-SIZE BASELINE configurations 2, 3, 4 and 5, from the CPU oracle.

    python tests/golden/make_golden_big.py [cfg2] [cfg2rough] [cfg3] [cfg4] [cfg5]      (default: all; ~10 min on 8 cores)

The brute-force oracle cannot render these configurations whole inside a test run (config 5 is 68.7 G
samples), so this script is run once in the build container and its outputs are committed:

  cfg2_full_1024x768x50_seed1.npz   SHA-256 of the WHOLE fp32 radiance and RGB8 image of config 2 (every one of the
                                    1024 columns through the oracle) + an 8x8 sub-sampled copy for diagnosis
  cfg2rough_cols_1024x768x50_seed1.npz every 32nd column of config 2 with the ROUGH stand-in (uneven triangle sizes,
                                    concavities, fins and spikes: rbrt_amd/standin.py make_rough_mesh) in the bunny's place
  cfg3_cols_1920x1080x512_seed1.npz every 64th column of config 3 (30 columns x 1080 rows x 512 spp)
  cfg4_windows_1024x768x50_seed1.npz config 4 (871,414-triangle stand-in): a window at the example scene's scale 45
                                    (where the reference's |a| < 1e-3 test makes every triangle invisible) and a window
                                    at scale 450 (visible), full 50 spp
  cfg5_tiles_4096x4096x4096_seed1.npz 13 8x8 pixel tiles of config 5 (scenes/header_card.yaml, 4096 spp): sky, ground,
                                    each of the six small spheres, five tiles on the mesh (centre, silhouette,
                                    ground contact). All tiles have tile_x % 8 == 5, i.e. they belong to rank 5 of an
                                    8-way sharding, so that one GPU's eighth of the frame covers them.

Fixtures are data (inputs by name + expected outputs). The GPU tests (tests/test_full_size.py) render every
configuration at full size through the C ABI and compare with these bit for bit; the not-gpu tests re-render a few
columns / one tile with the oracle so that an accidental edit of the checker is caught.
"""
import hashlib
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))
from oracle import pyoracle  # noqa: E402
import scenes  # noqa: E402
from rbrt_amd import abi, standin  # noqa: E402

SEED = 1
CFG3_STRIDE = 64
CFG2ROUGH_STRIDE = 32
# (what the tile centre's primary ray sees, tile_y, tile_x) in units of 8x8 tiles of the 4096x4096 frame
CFG5_TILES = [("sky", 103, 221), ("ground", 410, 253), ("sphere1", 308, 53), ("sphere2", 334, 341),
              ("sphere3", 264, 165), ("sphere4", 219, 429), ("sphere5", 339, 205), ("sphere6", 311, 453),
              ("mesh", 307, 309), ("mesh", 246, 389), ("mesh", 264, 253), ("mesh", 238, 301), ("mesh", 306, 309)]
# config 4: (mesh scale, translation, window c0 c1 r0 r1)
CFG4_CASES = {
    "scale45": dict(scale=45.0, translation=(5.0, -1.8, -12.5), window=(700, 764, 320, 352)),
    "scale450": dict(scale=450.0, translation=(50.0, -18.0, -145.0), window=(640, 736, 120, 168)),
}


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cfg4_scene(case):
    return scenes.example_scene(pyoracle, standin.DRAGON_TRIANGLES,
                                mesh_over={"scale": case["scale"], "translation": case["translation"]})


def make_cfg2():
    cam = scenes.camera(pyoracle, 1024, 768)
    sc = scenes.example_scene(pyoracle)
    t0 = time.time()
    rad, rgb, rays = pyoracle.render(cam, sc, abi.default_opts(spp=50, seed=SEED))
    np.savez_compressed(HERE / "cfg2_full_1024x768x50_seed1.npz", radiance_sha256=sha(rad), rgb8_sha256=sha(rgb),
                        radiance_sub=rad[::8, ::8].copy(), rgb8_sub=rgb[::8, ::8].copy(), rays=np.int64(rays),
                        mean=rad.mean(axis=(0, 1)), triangles=np.int64(standin.BUNNY_TRIANGLES))
    print(f"cfg2 {sha(rad)[:16]} rays {rays} ({time.time() - t0:.0f} s)", flush=True)


def make_cfg2rough():
    cam = scenes.camera(pyoracle, 1024, 768)
    sc = scenes.example_scene(pyoracle, kind="rough")
    t0 = time.time()
    rad, _, rays = pyoracle.render(cam, sc, abi.default_opts(spp=50, seed=SEED), want_rgb8=False, col_stride=CFG2ROUGH_STRIDE)
    cols = np.arange(0, 1024, CFG2ROUGH_STRIDE)
    np.savez_compressed(HERE / "cfg2rough_cols_1024x768x50_seed1.npz", cols=cols, radiance=rad[:, cols].copy(), rays=np.int64(rays))
    print(f"cfg2rough {len(cols)} columns, rays {rays} ({time.time() - t0:.0f} s)", flush=True)


def make_cfg3():
    cam = scenes.camera(pyoracle, 1920, 1080)
    sc = scenes.example_scene(pyoracle)
    t0 = time.time()
    rad, _, rays = pyoracle.render(cam, sc, abi.default_opts(spp=512, seed=SEED), want_rgb8=False, col_stride=CFG3_STRIDE)
    cols = np.arange(0, 1920, CFG3_STRIDE)
    np.savez_compressed(HERE / "cfg3_cols_1920x1080x512_seed1.npz", cols=cols, radiance=rad[:, cols].copy(),
                        rays=np.int64(rays))
    print(f"cfg3 {len(cols)} columns, rays {rays} ({time.time() - t0:.0f} s)", flush=True)


def make_cfg4():
    out = {}
    cam = scenes.camera(pyoracle, 1024, 768)
    for name, case in CFG4_CASES.items():
        t0 = time.time()
        c0, c1, r0, r1 = case["window"]
        rad, _, rays = pyoracle.render(cam, cfg4_scene(case), abi.default_opts(spp=50, seed=SEED), window=case["window"],
                                       want_rgb8=False)
        out[f"{name}_radiance"] = rad[r0:r1, c0:c1].copy()
        out[f"{name}_window"] = np.int64(case["window"])
        print(f"cfg4 {name} window {case['window']} rays {rays} mean {rad[r0:r1, c0:c1].mean():.4f} ({time.time() - t0:.0f} s)",
              flush=True)
    np.savez_compressed(HERE / "cfg4_windows_1024x768x50_seed1.npz", **out)


def make_cfg5():
    cam = scenes.camera(pyoracle, 4096, 4096)
    sc = scenes.header_scene(pyoracle)
    tiles = np.int64([(ty, tx) for _, ty, tx in CFG5_TILES])
    rad_tiles = np.zeros((len(CFG5_TILES), 8, 8, 3), np.float32)
    for k, (what, ty, tx) in enumerate(CFG5_TILES):
        t0 = time.time()
        win = (tx * 8, tx * 8 + 8, ty * 8, ty * 8 + 8)
        rad, _, rays = pyoracle.render(cam, sc, abi.default_opts(spp=4096, seed=SEED), window=win, want_rgb8=False)
        rad_tiles[k] = rad[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
        print(f"cfg5 tile {k} {what} ({ty},{tx}) rays {rays} mean {rad_tiles[k].mean():.4f} ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(HERE / "cfg5_tiles_4096x4096x4096_seed1.npz", tiles=tiles, radiance=rad_tiles,
                        what=np.array([w for w, _, _ in CFG5_TILES]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg5", "cfg4", "cfg3", "cfg2rough", "cfg2"]
    for w in which:
        {"cfg2": make_cfg2, "cfg2rough": make_cfg2rough, "cfg3": make_cfg3, "cfg4": make_cfg4, "cfg5": make_cfg5}[w]()
