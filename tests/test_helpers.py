"""Elastic launches (api.cpp): a launch that is already running can be given more waves -- a HELPER launch of the same kernel
that draws from the same work counters into the same sample buffer -- when the GPU has room and the caller has stopped
issuing. The reference's loop has nothing of the kind (rayon steals work inside one call, lib.rs:84-88); what must hold
is that the image does not change: every sample written exactly once before the resolve reads it, nothing drawn from
counters that have passed to the lane's next launch. RBRT_HELPERS=2 (lab) gives EVERY overlapped launch a helper, which
puts the protocol under every kind of launch; poisoned sample buffers (RBRT_POISON_SAMPLES=1) turn a sample that was not
written into NaN."""
import time

import numpy as np
import pytest

import scenes
from rbrt_amd import abi

pytestmark = pytest.mark.gpu


def _stream(hip, torch, sc, cams, opts_list, sizes):
    imgs = [torch.empty((h, w, 3), dtype=torch.float32, device="cuda") for (w, h) in sizes]
    with hip.HipScene(sc) as hs:
        hs.refine_wait(60.0)
        hs.set_timing(True)
        for cam, opts, img in zip(cams, opts_list, imgs):
            hs.render_device(cam, opts, img.data_ptr())
        torch.cuda.synchronize()
        n_helpers = hs.helper_launches()
        hs.check()  # (a helper count that never returned to zero would have raised the error flag)
    return [i.cpu().numpy() for i in imgs], n_helpers


@pytest.mark.parametrize("n_tris,size", [(3001, (160, 120)), (20000, (320, 240)), (0, (64, 48))])
def test_a_helper_with_every_launch_changes_no_frame(hip, oracle, monkeypatch, n_tris, size):
    import torch
    sc = scenes.example_scene(oracle, n_tris) if n_tris else abi.SceneData(spheres=scenes.EXAMPLE_SPHERES)
    w, h = size
    n = 24
    cams = [scenes.camera(oracle, w, h) for _ in range(n)]
    opts = [abi.default_opts(spp=3 + (k % 3), seed=10 + k) for k in range(n)]  # (every frame its own seed and sample count)
    monkeypatch.setenv("RBRT_HIP_LAB", "1")
    monkeypatch.setenv("RBRT_POISON_SAMPLES", "1")
    monkeypatch.setenv("RBRT_HELPERS", "0")
    ref, n0 = _stream(hip, torch, sc, cams, opts, [size] * n)
    assert n0 == 0
    monkeypatch.setenv("RBRT_HELPERS", "2")
    got, n2 = _stream(hip, torch, sc, cams, opts, [size] * n)
    assert n2 >= n - 8, n2  # (all but the launches that found the GPU idle)
    for k in range(n):
        assert np.isfinite(got[k]).all() and np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k
    exp, _, _ = oracle.render(cams[5], sc, opts[5])
    assert np.array_equal(ref[5].view(np.uint32), exp.view(np.uint32))


def test_helpers_with_sample_batches_ranks_and_changing_sizes(hip, oracle, monkeypatch):
    """Several launches per call (a small workspace), tiles of one rank of three, image sizes that change from call to call
    (sample buffers are made anew: every stream, the helpers' too, is drained first)."""
    import torch
    sc = scenes.example_scene(oracle, 3001)
    monkeypatch.setenv("RBRT_HIP_LAB", "1")
    monkeypatch.setenv("RBRT_POISON_SAMPLES", "1")
    monkeypatch.setenv("RBRT_HIP_WORKSPACE_MB", "1")
    sizes = [(96, 64), (160, 120), (96, 64), (200, 150), (160, 120), (96, 64)] * 2
    cams = [scenes.camera(oracle, w, h) for (w, h) in sizes]
    opts = [abi.default_opts(spp=9, seed=3 + k, tile_rank=k % 3, tile_world=3) for k in range(len(sizes))]
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("RBRT_HELPERS", mode)
        out[mode], _ = _stream(hip, torch, sc, cams, opts, sizes)
    for k, (w, h) in enumerate(sizes):
        n_loc = hip.packed_pixels(w, h, k % 3, 3) * 3
        a, b = out["0"][k].reshape(-1)[:n_loc], out["2"][k].reshape(-1)[:n_loc]
        assert np.isfinite(b).all() and np.array_equal(a.view(np.uint32), b.view(np.uint32)), k


def test_the_watcher_gives_the_end_of_a_stream_more_waves(hip, oracle, monkeypatch):
    """The automatic mode: frames issued back to back, then nothing: the last launches are given the slots the earlier ones
    free (helper launches > 0), and every frame is what it is without them."""
    import torch
    sc = scenes.example_scene(oracle, 20000)
    w, h, n = 1024, 768, 16  # (launches of 16 M work items or more are the ones the watcher helps: 1024 x 768 x 24)
    cams = [scenes.camera(oracle, w, h) for _ in range(n)]
    opts = [abi.default_opts(spp=24, seed=20 + k) for k in range(n)]
    monkeypatch.setenv("RBRT_HIP_LAB", "1")
    monkeypatch.setenv("RBRT_POISON_SAMPLES", "1")
    monkeypatch.setenv("RBRT_HELPERS", "0")
    ref, _ = _stream(hip, torch, sc, cams, opts, [(w, h)] * n)
    monkeypatch.delenv("RBRT_HELPERS")
    total = 0
    for _ in range(6):  # (up to six streams, each with its own end: whether the watcher gets to help one depends on the box's load)
        got, n_helpers = _stream(hip, torch, sc, cams, opts, [(w, h)] * n)
        total += n_helpers
        for k in range(n):
            assert np.isfinite(got[k]).all() and np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k
        time.sleep(0.01)
        if total > 0 and _ >= 2:
            break
    assert total > 0
