"""SURVEY 8(c) fixture 4: single scatter events -- material + incoming ray + hit point + hit normal + an explicit random
stream -> (bool, attenuation, next ray, stream state afterwards). 300 events for each of five materials
(tests/golden/scatter_events.npz, made by tests/golden/make_golden.py from the oracle).

not gpu: the oracle still reproduces the fixture, and so does the independent numpy restatement (np_reference).
gpu    : the device functions the megakernel shades with (rbrt_hip_debug_scatter) give the same direction, bool and
         stream state, bit for bit -- RayScattering::scatter of lambertian.rs:11-24, metal.rs:12-25, dielectric.rs:11-85 with
         materials.rs:14-37 underneath. An image mismatch can be localised to one material with this, without bisecting images.
"""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import np_reference as R
from rbrt_amd import abi

GOLD = Path(__file__).resolve().parent / "golden"


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ev():
    return np.load(GOLD / "scatter_events.npz")


def test_fixture_covers_every_material_and_both_outcomes(ev):
    kinds = set(zip(ev["kind"].tolist(), np.round(ev["param"].astype(np.float64), 3).tolist()))
    assert kinds == {(0, 0.0), (1, 0.005), (1, 0.6), (2, 1.8), (2, 0.2)}
    assert len(ev["kind"]) == 1500
    metal = ev["kind"] == 1
    assert 0 < (ev["ok"][metal] == 0).sum() < metal.sum()         # metal.rs:25 returns false sometimes
    assert (ev["ok"][~metal] == 1).all()                            # lambertian.rs:23, dielectric.rs:58: always true
    assert np.array_equal(ev["attenuation"][ev["kind"] == 2], np.ones(((ev["kind"] == 2).sum(), 3), np.float32))  # dielectric.rs:18


def test_oracle_and_numpy_restatement_reproduce_the_fixture(oracle, ev):
    u32p = C.POINTER(C.c_uint32)
    for i in range(0, len(ev["kind"]), 7):  # every 7th event: 215 of them
        mat = abi.material(int(ev["kind"][i]), tuple(float(x) for x in ev["albedo"][i]), float(ev["param"][i]))
        att, out = np.zeros(3, np.float32), np.zeros(6, np.float32)
        b, a = np.zeros(2, np.uint32), np.zeros(2, np.uint32)
        seed, pixel, sample = (int(x) for x in ev["stream_key"][i])
        ok = oracle.lib().rbrt_oracle_kat_scatter_state(C.byref(mat), oracle._p(ev["in_ray"][i].copy()), oracle._p(ev["point"][i].copy()),
                                                        oracle._p(ev["normal"][i].copy()), seed, pixel, sample, oracle._p(att), oracle._p(out),
                                                        b.ctypes.data_as(u32p), a.ctypes.data_as(u32p))
        assert ok == ev["ok"][i] and np.array_equal(bits(att), bits(ev["attenuation"][i])) and np.array_equal(bits(out), bits(ev["out_ray"][i]))
        assert np.array_equal(b, ev["state_before"][i]) and np.array_equal(a, ev["state_after"][i])
        r = R.Rng(seed, pixel, sample)
        assert (r.s0, r.s1) == tuple(int(x) for x in ev["state_before"][i])
        got_ok, got_att, no, nd = R.scatter((int(ev["kind"][i]), ev["albedo"][i], np.float32(ev["param"][i])), ev["in_ray"][i][3:],
                                            dict(point=ev["point"][i], normal=ev["normal"][i]), r)
        assert got_ok == bool(ev["ok"][i]) and (r.s0, r.s1) == tuple(int(x) for x in ev["state_after"][i]), i
        assert np.array_equal(bits(got_att), bits(ev["attenuation"][i])), i
        assert np.array_equal(bits(no), bits(ev["out_ray"][i][:3])) and np.array_equal(bits(nd), bits(ev["out_ray"][i][3:])), i


@pytest.mark.gpu
def test_device_scatter_functions_replay_the_fixture(hip, ev):
    out_dir, ok, st = hip.debug_scatter(ev["kind"], ev["albedo"], ev["param"], ev["in_ray"][:, 3:], ev["point"], ev["normal"], ev["state_before"])
    assert np.array_equal(ok, ev["ok"])
    assert np.array_equal(st, ev["state_after"]), "draw count / order differs"
    good = ev["ok"] == 1  # (the direction of a failed metal scatter is computed too; compare all the same)
    bad = (bits(out_dir) != bits(ev["out_ray"][:, 3:])).any(axis=1)
    assert not bad.any(), (np.nonzero(bad)[0][:8].tolist(), ev["kind"][bad][:8].tolist(), int(good.sum()))
