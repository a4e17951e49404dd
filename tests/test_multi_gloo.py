"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups run bench.py's exchange step — every
rank packs ITS pixel tiles of a known image, one gather to rank 0, rank 0 de-interleaves — and the
result must equal the image. The per-rank render itself needs a GPU and is covered by
tests/test_gpu_parity.py::test_tile_sharding_is_partition_invariant."""
import os
import socket

import numpy as np
import pytest

from rbrt_amd import tiles


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)  # same image on every rank
    image = rng.random((h, w, 3), dtype=np.float32)
    sizes = [tiles.packed_pixels(w, h, r, world) * 3 for r in range(world)]
    maxn = max(sizes)
    mine = torch.zeros(maxn, dtype=torch.float32)
    mine[:sizes[rank]] = torch.from_numpy(tiles.pack(image, rank, world).reshape(-1))
    gathered = [torch.zeros(maxn, dtype=torch.float32) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)          # same call shape as bench.py (equal-size, padded tail)
    ok = True
    if rank == 0:
        parts = [gathered[r][:sizes[r]].numpy().reshape(-1, 3) for r in range(world)]
        ok = bool(np.array_equal(tiles.unpack(parts, w, h), image))
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py's max-over-ranks timing reduction
    ok = ok and t.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("world,w,h", [(2, 100, 60), (2, 1024, 768), (3, 33, 9)])
def test_gather_and_unpack_over_gloo(world, w, h):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _counter_worker(rank, world, port, q):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    set_flag, wait_flag, err = bench.open_step_counters(dist, np, rank, world, lambda fn: fn(), timeout_s=20.0)
    ok = err == ""
    lag = 3  # (bench.py's NBUF - 1: a peer may run this many steps ahead of rank 0's consumption)
    for k in range(1, 201):
        if rank != 0 and k > lag:
            wait_flag(world, k - lag)  # rank 0 has consumed the step whose buffer this one reuses
        set_flag(rank, k)              # "my record for step k has been issued"
        if rank == 0:
            for r in range(1, world):
                wait_flag(r, k)
            set_flag(world, k)
    dist.barrier()
    try:  # a counter nobody advances: the wait ends with an error, not a hang
        import time
        t0 = time.perf_counter()
        short = bench.open_step_counters(dist, np, rank, world, lambda fn: fn(), timeout_s=0.3)[1]
        try:
            short(world, 10 ** 9)
            ok = False
        except RuntimeError as e:
            ok = ok and "waited" in str(e) and time.perf_counter() - t0 < 10.0
    finally:
        dist.barrier()
        dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("world", [2, 3])
def test_step_counters_of_the_ipc_gather(world):
    """bench.py --gather ipc orders its waits on interprocess events through step counters in /dev/shm
    (bench.open_step_counters): rank 0 waits on the host until every peer has issued step k, a peer until rank 0 has
    consumed the step whose buffer it is about to reuse; a counter that never advances raises instead of hanging."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_counter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_pipeline_note_follows_the_hardware_queues(monkeypatch):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert bench.pipeline_note(0).startswith("auto: 8 trace launches in flight, 8 side by side on 3 (4 for launches of 8 M")
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    assert bench.pipeline_note(0).startswith("auto: 4 trace launches in flight, 4 side by side on 6 of a CU")
    assert bench.pipeline_note(1) == "1 (no overlap between steps)"


def test_pack_unpack_roundtrip_and_partition():
    rng = np.random.default_rng(1)
    for (w, h) in ((1, 1), (9, 17), (100, 60), (64, 64)):
        img = rng.random((h, w, 3), dtype=np.float32)
        for world in (1, 2, 3, 8):
            parts = [tiles.pack(img, r, world) for r in range(world)]
            assert np.array_equal(tiles.unpack(parts, w, h), img)
            assert sum(len(p) for p in parts) == tiles.n_tiles(w, h) * 64


def test_bench_gpus_n_by_itself_starts_ranks_as_a_child_and_passes_their_exit_code_on():
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks under torch.distributed.run as a child
    process (it never touches the GPU itself) and exits with their code. Without a GPU the ranks end with bench.py's own
    message -- which shows they were started, with WORLD_SIZE=2, and that the failure reaches the caller."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU form of this test is tests/test_multi_rank_one_gpu.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-single-gpu", "--steps", "1",
                        "--warmup", "0", "--cpu-col-stride", "0"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode != 0
    assert "as a child process" in r.stderr and r.stderr.count("bench.py needs a GPU") >= 1, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # a WORLD_SIZE that contradicts --gpus is still refused, not relaunched
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=60,
                       cwd=root, env=dict(env, WORLD_SIZE="1"))
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
