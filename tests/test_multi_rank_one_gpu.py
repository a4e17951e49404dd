"""The N-rank code paths, executed on the ONE GPU a test box has.

What the reference does here is one rayon loop over columns (rbrt_lib/src/lib.rs:84-113); the replacement shards 8x8 pixel
tiles over ranks. Two hosts drive it and both are run here with more ranks than GPUs:

* the C++ host `rbrt --gpus N --oversubscribe` (rank r on device r % n_devices): worker threads, per-pass barriers,
  per-rank checkpoint slots, an interrupted + resumed run, the host-side merge of the ranks' packed tiles;
* `bench.py --gpus N --rehearse-single-gpu` under torch.distributed.run: N processes on cuda:0, the side-stream gather
  with gloo standing in for RCCL, and the IPC gather (rank 0 copying out of the peers' device buffers) as it is.

What can NOT run on one GPU is RCCL itself (a communicator refuses two ranks on one device): `--gather rccl` with
duplicate devices must fail with a message, not hang; its code is kept as small as it is (render.cpp).
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import scenes
from rbrt_amd import abi, standin

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
RBRT = ROOT / "rbrt_amd" / "bin" / "rbrt"
W, H = 136, 100  # ragged: 17 x 13 tiles, the last tile row is half empty


def _scene_file(tmp_path, n_tris=1203):
    v, f = standin.make_mesh(n_tris)
    standin.write_obj(tmp_path / "bunny.obj", v, f)
    text = (ROOT / "scenes" / "example_scene.yaml").read_text().replace("bunny.obj", str(tmp_path / "bunny.obj"))
    (tmp_path / "scene.yaml").write_text(text)
    return tmp_path / "scene.yaml"


def _png(path):
    from PIL import Image
    return np.array(Image.open(path))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_cli_n_ranks_on_one_gpu_checkpoint_resume_and_host_merge(hip, oracle, tmp_path, world):
    """`rbrt --gpus N --oversubscribe` in passes with a checkpoint, interrupted after the second pass and resumed:
    every rank's running sums go through the checkpoint file, the ranks meet at the per-pass barriers, and the host
    merge of their packed tiles gives the oracle's image bit for bit (8-bit PNG)."""
    cfg = _scene_file(tmp_path)
    out, ck, rep = tmp_path / "out.png", tmp_path / "render.ckpt", tmp_path / "report.json"
    cmd = [str(RBRT), "-c", str(cfg), "-t", str(out), "--height", str(H), "-w", str(W), "-s", "10", "--seed", "6",
           "--gpus", str(world), "--oversubscribe", "--gather", "host", "--pass-samples", "3", "--checkpoint", str(ck),
           "--report", str(rep)]
    r1 = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, RBRT_TEST_STOP_AFTER_PASS="2"))
    assert r1.returncode == 101 and "stopped after pass 2" in r1.stderr, (r1.returncode, r1.stderr[-2000:])
    assert ck.exists() and not out.exists() and not rep.exists()
    assert "Rendering 30.0% complete!" in r1.stdout and "Rendering 60.0% complete!" in r1.stdout
    # header (48 B) + per rank: u64 count + its packed running sums
    expect = 48 + sum(8 + hip.packed_pixels(W, H, r, world) * 3 * 4 for r in range(world))
    assert ck.stat().st_size == expect
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert "Resuming from checkpoint" in r2.stdout and "at sample 6 of 10" in r2.stdout
    assert "Rendering 90.0% complete!" in r2.stdout and "Rendering 100% complete!" in r2.stdout
    assert not ck.exists()
    _, exp8, _ = oracle.render(scenes.camera(oracle, W, H), scenes.example_scene(oracle, 1203), abi.default_opts(spp=10, seed=6))
    assert np.array_equal(_png(out), exp8)
    j = json.loads(rep.read_text())
    assert (j["gpus"], j["gather"], j["width"], j["height"], j["samples"], j["seed"]) == (world, "host", W, H, 10, 6)
    assert j["resumed_from_sample"] == 6 and j["passes"] == 2 and j["triangles"] == 1203 and j["bvh_builder"] in ("host", "device")
    assert j["render_s"] > 0 and j["upload_build_s"] > 0 and j["mray_samples_per_s"] > 0
    # where the run's time went: every part named, and the parts add up to the total (other_s is what none of them covers)
    parts = ("parse_s", "obj_load_s", "prep_s", "hip_init_s", "upload_s", "bvh_build_s", "lanes_s", "buffers_s", "render_s", "gather_s",
             "release_s", "encode_s", "other_s")
    for k in parts + ("total_s",):
        assert j[k] >= 0, k
    # (with several ranks every phase is its SLOWEST rank's: the parts can add up to a little more than the wall clock)
    assert 0.9 * j["total_s"] <= sum(j[k] for k in parts) <= 1.5 * j["total_s"] and j["obj_load_s"] > 0 and j["hip_init_s"] > 0


def test_cli_one_pass_n_ranks_equals_one_rank(hip, tmp_path):
    """No passes, no checkpoint (one barrier-free pass per rank): the 5-rank image is the 1-rank image, byte for byte."""
    cfg = _scene_file(tmp_path, 2004)
    outs = []
    for world in (1, 5):
        out = tmp_path / f"w{world}.png"
        r = subprocess.run([str(RBRT), "-c", str(cfg), "-t", str(out), "--height", str(H), "-w", str(W), "-s", "4", "--gpus", str(world),
                            "--oversubscribe"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(_png(out))
    assert np.array_equal(outs[0], outs[1])


def test_cli_rccl_gather_falls_back_to_the_host_gather_instead_of_failing(hip, oracle, tmp_path):
    """A communicator cannot hold two ranks on one device, and --gather rccl is a request, not a condition: whatever keeps
    RCCL from doing the gather (here: shared devices; on a real node: a missing library, communicators that do not come
    up, a failed exchange) sends the ranks to the host gather IN THE SAME PROCESS -- never a hang, never a second start of
    a process that has touched the GPU -- and the run says so on stderr and in --report. The image is the oracle's."""
    if hip.device_count() >= 2:
        pytest.skip("more than one GPU: RCCL can run here")
    cfg = _scene_file(tmp_path)
    rep, out = tmp_path / "rep.json", tmp_path / "o.png"
    r = subprocess.run([str(RBRT), "-c", str(cfg), "-t", str(out), "--height", "48", "-w", "64", "-s", "2", "--gpus", "2",
                        "--oversubscribe", "--gather", "rccl", "--report", str(rep), "--seed", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "one GPU per rank" in r.stderr and "host memory instead" in r.stderr
    j = json.loads(rep.read_text())
    assert j["gather"].startswith("host (rccl was asked for") and j["gpus"] == 2
    _, exp8, _ = oracle.render(scenes.camera(oracle, 64, 48), scenes.example_scene(oracle, 1203), abi.default_opts(spp=2, seed=3))
    assert np.array_equal(_png(out), exp8)
    # without --oversubscribe the old refusal stands
    r = subprocess.run([str(RBRT), "-c", str(cfg), "-t", str(tmp_path / "o.png"), "--height", "48", "-w", "64", "-s", "2", "--gpus", "2"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 101 and "requested 2 GPUs, 1 present" in r.stderr
    # one GPU asked to gather over RCCL has nothing to gather: it must not even load the library
    r = subprocess.run([str(RBRT), "-c", str(cfg), "-t", str(tmp_path / "o.png"), "--height", "48", "-w", "64", "-s", "2", "--gather", "rccl"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(extra, launcher=(), env=None, expect_failure=False):
    cmd = [sys.executable, *launcher, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--width", "256", "--height", "192", "--spp", "6",
           "--triangles", "3003", "--cpu-col-stride", "0", "--isolated-steps", "0", "--single-frames", "0", "--vary-seed", "1", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT), env=dict(os.environ, **(env or {})))
    if expect_failure:
        assert r.returncode != 0, r.stdout[-1500:]
        return r
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("world,gather", [(2, "rccl"), (3, "ipc"), (3, "auto")])
def test_bench_sharded_step_on_one_gpu_gives_the_single_gpu_frame(hip, world, gather):
    """bench.py's N > 1 step -- tiles of rank r, two buffer sets, gather on a side stream behind an event, strided
    unpack on rank 0 -- as `world` fresh processes on cuda:0 (gloo through host memory in place of RCCL): the last
    frame's SHA-256 equals the single-process run's. Seeds vary per step, so a stale buffer would show."""
    one = _bench(["--gpus", "1"])
    many = _bench(["--gpus", str(world), "--rehearse-single-gpu", "--gather", gather, "--gather-probe-steps", "3"],
                  launcher=("-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port())))
    assert many["n_gpus"] == world and one["n_gpus"] == 1
    assert many["config"]["image_sha256_16"] == one["config"]["image_sha256_16"]
    # what an 8-GPU run will print to diagnose itself with: the per-rank split of a step and the
    # collective's own account of its ranks
    ph, co = many["phases"], many["collective"]
    assert [p["rank"] for p in ph["per_rank"]] == list(range(world))
    for p in ph["per_rank"]:
        assert all(p[k] >= 0.0 for k in ("render_ms", "gather_wait_ms", "gather_ms", "unpack_ms")) and p["render_ms"] > 0.0
    assert ph["max"]["render_ms"] == max(p["render_ms"] for p in ph["per_rank"])
    assert co["ranks"] == world and co["backend"] == "gloo" and co["gather_bytes_per_rank"] > 0
    # how rank 0 got the tiles: the backend's gather, or its own copies out of the peers' buffers (HIP IPC between the
    # processes, interprocess events, step counters in /dev/shm) -- the second is the same code on one GPU as on eight
    assert co["gather_asked"] == gather and co["ipc_gather_available"] == (gather != "rccl"), co
    if gather == "auto":
        assert set(co["gather_probe_ms_per_step"]) == {"rccl", "ipc"} and co["gather"].startswith(("dist.gather", "rank 0 copies"))
    else:
        assert co["gather"].startswith("dist.gather" if gather == "rccl" else "rank 0 copies") and co["gather_probe_ms_per_step"] is None
    for j in (one, many):  # every timed step renders a camera the library has not seen; the cached-camera leg is reported apart
        assert j["config"]["tile_pass"]["in_timed_region"] is True and j["ms_per_step_new_camera"] == j["ms_per_step"]
        assert j["ms_per_step_same_camera"] > 0.0 and j["same_camera_leg"]["steps"] == 3
    assert "phases" not in one


@pytest.mark.parametrize("stage", ["export:1", "open:0"])
def test_bench_ipc_gather_that_cannot_be_set_up_leaves_the_backends_gather(hip, stage):
    """The IPC gather's set-up ends with an agreement step: a failure on ANY rank, while exporting its buffers or while
    opening the peers', makes EVERY rank gather with dist.gather (what a node does whose driver refuses to map another
    device's memory) -- the frame is the single-GPU frame and the line says why. Asked for by name (--gather ipc) the
    same failure ends all ranks with a message and a non-zero exit code instead."""
    world = 3
    launcher = ("-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port")
    one = _bench(["--gpus", "1"])
    many = _bench(["--gpus", str(world), "--rehearse-single-gpu", "--gather", "auto"], launcher=(*launcher, str(_free_port())),
                  env={"RBRT_BENCH_IPC_FAIL": stage})
    co = many["collective"]
    assert many["config"]["image_sha256_16"] == one["config"]["image_sha256_16"]
    assert co["ipc_gather_available"] is False and co["gather"].startswith("dist.gather") and co["gather_probe_ms_per_step"] is None
    assert f"rank {stage.split(':')[1]}" in co["ipc_gather_note"] and "injected failure" in co["ipc_gather_note"]
    r = _bench(["--gpus", str(world), "--rehearse-single-gpu", "--gather", "ipc"], launcher=(*launcher, str(_free_port())),
               env={"RBRT_BENCH_IPC_FAIL": stage}, expect_failure=True)
    assert "--gather ipc cannot be set up" in r.stderr and "injected failure" in r.stderr


def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks(hip):
    """`python bench.py --gpus 2 ...` with no torch.distributed.run in front of it -- the form a driver may use for the
    scaling run -- starts the ranks itself as a child process and prints exactly ONE JSON line, from rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--rehearse-single-gpu", "--steps", "2", "--warmup", "1",
           "--cpu-col-stride", "0", "--width", "256", "--height", "192", "--spp", "6", "--triangles", "3003"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(ROOT), env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["collective"]["ranks"] == 2 and j["steps"] == 2 and j["warmup"] == 1
    assert "as a child process" in r.stderr
