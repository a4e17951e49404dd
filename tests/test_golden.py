"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).

not-gpu: the oracle still reproduces them (pins the checker against accidental edits).
gpu    : the HIP path reproduces them through the C ABI without the oracle in the loop.
"""
import hashlib
import sys
from pathlib import Path

import numpy as np
import pytest

import scenes
from rbrt_amd import abi

GOLD = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLD))
import make_golden  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check(name, rad, rgb):
    g = np.load(GOLD / f"{name}.npz")
    sub = rad[::4, ::4]
    assert np.array_equal(sub.view(np.uint32), g["radiance_sub"].view(np.uint32)), \
        f"{name}: sub-sampled radiance differs, max abs {np.abs(sub - g['radiance_sub']).max()}"
    assert sha(rad) == str(g["radiance_sha256"]), f"{name}: full radiance hash differs"
    assert sha(rgb) == str(g["rgb8_sha256"]), f"{name}: RGB8 hash differs"


@pytest.mark.parametrize("name", list(make_golden.CASES))
def test_oracle_reproduces_golden(oracle, name):
    case = make_golden.CASES[name]
    cam = scenes.camera(oracle, case["w"], case["h"])
    rad, rgb, rays = oracle.render(cam, make_golden.build_scene(case), abi.default_opts(spp=case["spp"], seed=case["seed"]))
    check(name, rad, rgb)
    assert rays == int(np.load(GOLD / f"{name}.npz")["rays"])


def test_oracle_reproduces_ray_records(oracle):
    g = np.load(GOLD / "rays_example3001.npz")
    t, obj, tri, dist = oracle.trace_rays(scenes.example_scene(oracle, 3001), g["rays"])
    assert np.array_equal(obj, g["obj"]) and np.array_equal(tri, g["tri"])
    assert np.array_equal(t.view(np.uint32), g["t"].view(np.uint32))
    assert np.array_equal(dist.view(np.uint32), g["dist"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(make_golden.CASES))
def test_hip_reproduces_golden(hip, oracle, name):
    case = make_golden.CASES[name]
    cam = scenes.camera(oracle, case["w"], case["h"])  # scene PREP only; the expected image is the fixture
    rad, rgb = hip.render_scene(cam, case["spp"], make_golden.build_scene(case), seed=case["seed"])
    check(name, rad, rgb)


@pytest.mark.gpu
def test_hip_reproduces_ray_records(hip, oracle):
    g = np.load(GOLD / "rays_example3001.npz")
    with hip.HipScene(scenes.example_scene(oracle, 3001)) as hs:
        t, obj, tri, dist = hs.trace_rays(g["rays"])
    assert np.array_equal(obj, g["obj"]) and np.array_equal(tri, g["tri"])
    assert np.array_equal(t.view(np.uint32), g["t"].view(np.uint32))
    assert np.array_equal(dist.view(np.uint32), g["dist"].view(np.uint32))
