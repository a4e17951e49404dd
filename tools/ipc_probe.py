#!/usr/bin/env python3
"""Probe (GPU box): device buffers and events shared between the ranks of a torch.distributed job through CUDA/HIP IPC,
rank 0 pulling every peer's buffer with a device copy -- the gather bench.py's `--gather ipc` does.
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 tools/ipc_probe.py
"""
import os
import sys
import time

import torch
import torch.distributed as dist
from torch.multiprocessing.reductions import reduce_tensor


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("PROBE_DEVICE", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    n = 1 << 18
    mine = torch.full((n,), float(rank + 1), dtype=torch.float32, device=dev)
    ev = torch.cuda.Event(enable_timing=False, interprocess=True)
    ev.record()
    payload = (reduce_tensor(mine), ev.ipc_handle())
    got = [None] * world
    dist.all_gather_object(got, payload)
    peers, peer_ev = [], []
    if rank == 0:
        for r in range(world):
            (fn, args), h = got[r]
            peers.append(mine if r == 0 else fn(*args))
            peer_ev.append(ev if r == 0 else torch.cuda.Event.from_ipc_handle(dev, h))
        print("rank 0 opened", [tuple(p.shape) for p in peers], flush=True)
    dist.barrier()
    out = torch.empty((world, n), dtype=torch.float32, device=dev) if rank == 0 else None
    for k in range(1, 6):
        mine.fill_(float(100 * k + rank))
        ev.record()
        dist.barrier()  # (stands in for the shared-memory step counters: the record is issued before the wait is)
        if rank == 0:
            s = torch.cuda.current_stream()
            for r in range(world):
                s.wait_event(peer_ev[r])
                out[r].copy_(peers[r], non_blocking=True)
            torch.cuda.synchronize()
            exp = [float(100 * k + r) for r in range(world)]
            have = [float(out[r, 0].item()) for r in range(world)] + [float(out[r, -1].item()) for r in range(world)]
            assert have == exp + exp, (k, have, exp)
        dist.barrier()  # (the peers may overwrite their buffers again)
    if rank == 0:
        print("ipc probe ok", flush=True)
    torch.cuda.synchronize()
    dist.barrier()
    del peers, peer_ev
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
