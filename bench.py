#!/usr/bin/env python3
"""bench.py — Mray-samples/s of the HIP path-tracing hot path on BASELINE config 2.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... --
     or by itself: with WORLD_SIZE unset, bench.py starts its N ranks under torch.distributed.run as a child process)

A step = one full render of the workload: scenes/example_scene.yaml (the reference's example scene)
with the 69,451-triangle STAND-IN for bunny.obj (the real asset is not available offline),
1024x768, 50 samples per pixel, seed 1. With N ranks the image's 8x8 pixel tiles are dealt
round-robin to the ranks (every rank holds the whole scene + BVH), each rank renders its tiles with
the same kernel, and a gather over xGMI brings the packed fp32 radiance to rank 0, which
de-interleaves it — so the total work per step is fixed ("scaling": "strong"). The gather is RCCL's (dist.gather) or
rank 0 pulling the peers' buffers through HIP IPC with device copies (--gather; by default a short untimed probe picks
the faster: RCCL's copy kernels need wave slots beside the persistent trace launches, the copy engines do not).

The scene is resident in HBM before the timed region (upload + BVH build are setup); the timed
region is K x [render kernels + gather + unpack], bracketed by barrier + synchronize, MAX over ranks.
EVERY step of the timed region renders a camera the library has not seen before (the position moves by one unit in the
last place per step): render_scene takes the camera per call (lib.rs:75-79), and the library's per-camera tile pass
(primary_cull_kernel + tile_lists_kernel) therefore runs inside the timed region, every step. A second leg renders one
camera over and over (a stream of frames of a fixed view, where that pass is paid once) and is reported beside it as
`ms_per_step_same_camera`; it is not `value`.

Rank 0 prints ONE JSON line with, besides the contract's fields:
  roofline     — the trace kernel's ALGORITHMIC bytes / the duration of an ISOLATED launch (a short second leg with
                 the frame pipeline off, so that a launch's HIP-event span is the cost of its own work and
                 kernel_ms <= that leg's ms per step) vs 8 TB/s HBM; beside it the figures that describe the
                 kernel as it really runs: measured_hbm (rocprofv3 PMC traffic, from profiles/), issue_mix (the
                 aggregate instruction-issue bound: the tree lives in L1/L2, the kernel is issue-bound) and valu_issue
                 (full-rate VALU alone) -- reported only from a profile measured on the same kernel sources
  single_frame — one blocking frame including the device-to-host copy of the RGB8 image (what a caller of the
                 reference's blocking render_scene would see); `value` is steady-state throughput with consecutive
                 frames overlapping in the library's frame pipeline
  cpu_baseline — the CPU oracle (a C++ restatement of the reference's AVX path; the Rust reference
                 cannot be built here) timed on this host on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# The library overlaps up to eight trace launches on streams of its own; the HIP runtime maps streams onto FOUR hardware
# queues unless told otherwise, and lanes that share a queue wait for each other (depth_for; an
# integrator exports the same). It has to be in the environment before the runtime starts, i.e. before
# torch is imported; a value the user exported is left alone.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
TRI_ALG_BYTES = 36     # the 9 fp32 SoA values Moller-Trumbore reads per triangle (triangle.rs:177-187)
NORMAL_ALG_BYTES = 12  # normal fetch per accepted mesh hit (mesh.rs:253-257)
PIXEL_ALG_BYTES = 12   # fp32 radiance store per pixel
# VALU issue, measured on this chip (profiles/r03_valu_rate.txt): SIMD cycles per wave64 instruction of plain f32 / u32
# VALU with >= 4 resident waves per SIMD, and what ONE wave alone sustains.
VALU_CYCLES_PER_INST = 2.0
VALU_CYCLES_ONE_WAVE = 4.5
CLOCK_GHZ = 2.4
# The same file prices the other instruction classes (SIMD cycles each, four waves resident): VALU that is not
# fma/add/mul/sub/add_u32/xor (compares, selects, min/max, integer multiplies, conversions, lane ops) ~3.1, transcendental
# 6.1, a scalar instruction ~2.0 (it takes an issue slot of its own), an LDS instruction ~6, a vector-memory one ~4. The
# kernel's static VALU mix (tools/static_cost.py: 62 % full-rate, 35 % of the 3.1 kind, 2 % transcendental) applied to the
# counted instructions gives the AGGREGATE issue bound reported as roofline.issue_mix -- the bound this kernel runs
# against (calibration: 100 extra v_add_f32 per traversal step cost 0.47 ms = 2.15 cycles each).
INT_FULL_FRAC = 0.58  # of the binary's integer VALU instructions, those that issue at the full rate (add / sub / xor / and / or / shifts)
ISSUE_COST = {"valu_mean": 0.62 * 2.0 + 0.355 * 3.1 + 0.025 * 6.1, "salu": 2.0, "lds": 6.0, "vmem": 4.0}


def pipeline_note(asked):
    """What the library does with the frame pipeline (api.cpp depth_for / grid_for), for the JSON line."""
    if asked == 1:
        return "1 (no overlap between steps)"
    try:
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        queues = 4
    depth = asked or (8 if queues >= 8 else 4)
    side = max(2, min(depth, queues))
    per_cu = -(-24 // side)
    return (f"{asked or f'auto: {depth}'} trace launches in flight, {min(depth, queues)} side by side on {per_cu}{' (4 for launches of 8 M work items or more)' if per_cu < 4 else ''} "
            f"of a CU's 16 wave slots each while they overlap; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")


def open_step_counters(dist, np, rank, world, guarded, timeout_s, failed=False):
    """world + 1 step counters shared by the ranks of one node through a file in /dev/shm (rank 0 makes it, the others open
    it, rank 0 unlinks it once everybody has it open). Collective. Returns (set_flag(i, v), wait_flag(i, v), error text):
    counter i is written by one rank only (rank i its own, rank 0 also counter `world`); wait_flag spins on the host until
    counter i has reached v and raises after timeout_s."""
    err, flags = "", None
    name = [f"/dev/shm/rbrt_bench_flags_{os.getpid()}_{os.environ.get('MASTER_PORT', '0')}"]
    try:
        if rank == 0 and not failed:
            flags = np.memmap(name[0], dtype=np.int64, mode="w+", shape=(world + 1,))
            flags[:] = 0
            flags.flush()
    except Exception as e:
        err = f"flags: {type(e).__name__}: {e}"
    guarded(lambda: dist.broadcast_object_list(name, src=0))
    try:
        if rank != 0 and not failed:
            flags = np.memmap(name[0], dtype=np.int64, mode="r+", shape=(world + 1,))
    except Exception as e:
        err = f"flags: {type(e).__name__}: {e}"
    guarded(lambda: dist.barrier())  # (every rank has the file open before rank 0 unlinks it)
    if rank == 0:
        try:
            os.unlink(name[0])
        except OSError:
            pass

    def set_flag(i, v):
        flags[i] = v

    def wait_flag(i, v):
        t0 = time.perf_counter()
        while int(flags[i]) < v:
            if time.perf_counter() - t0 > timeout_s:
                raise RuntimeError(f"IPC gather: rank {rank} waited {timeout_s:.0f} s for step {v} of counter {i} (at {int(flags[i])})")
            time.sleep(0)

    return set_flag, wait_flag, err


def setup_ipc_gather(torch, dist, np, rank, world, dev, mine, nbuf, guarded, timeout_s=60.0):
    """The IPC gather's shared objects (see main(): "The IPC gather"). Collective: every rank calls it. Returns a dict with
    ok / why and, when ok, the opened buffers and events, and set_flag / wait_flag over the ranks' step counters in /dev/shm.
    Any failure on any rank makes every rank return ok = False (the caller then gathers with dist.gather)."""
    st = {"ok": False, "why": "", "last_use": [0] * nbuf}
    payload, err = None, ""
    inject = os.environ.get("RBRT_BENCH_IPC_FAIL", "")  # (tests: "export:1" / "open:0" = that stage fails on that rank)
    try:
        if inject == f"export:{rank}":
            raise RuntimeError("injected failure")
        from torch.multiprocessing.reductions import reduce_tensor
        rendered = [torch.cuda.Event(enable_timing=False, interprocess=True) for _ in range(nbuf)]
        consumed = [torch.cuda.Event(enable_timing=False, interprocess=True) for _ in range(nbuf)] if rank == 0 else []
        for e in rendered + consumed:
            e.record()
        payload = {"mine": [reduce_tensor(t) for t in mine], "rendered": [e.ipc_handle() for e in rendered],
                   "consumed": [e.ipc_handle() for e in consumed]}
        st["rendered"], st["consumed"] = rendered, consumed
    except Exception as e:  # (a torch build or a driver without IPC)
        err = f"export: {type(e).__name__}: {e}"
    got = [None] * world
    guarded(lambda: dist.all_gather_object(got, (payload, err)))
    if not err and all(g[0] is not None for g in got):
        try:
            if inject == f"open:{rank}":
                raise RuntimeError("injected failure")
            if rank == 0:
                st["peer_mine"] = [None] + [[fn(*a) for fn, a in got[r][0]["mine"]] for r in range(1, world)]
                st["peer_rendered"] = [None] + [[torch.cuda.Event.from_ipc_handle(dev, h) for h in got[r][0]["rendered"]]
                                                for r in range(1, world)]
            else:
                st["consumed_by_rank0"] = [torch.cuda.Event.from_ipc_handle(dev, h) for h in got[0][0]["consumed"]]
        except Exception as e:
            err = f"open: {type(e).__name__}: {e}"
    set_flag, wait_flag, flags_err = open_step_counters(dist, np, rank, world, guarded, timeout_s, failed=bool(err))
    err = err or flags_err
    errs = [None] * world
    guarded(lambda: dist.all_gather_object(errs, err))
    bad = [f"rank {r}: {e or g[1]}" for r, (e, g) in enumerate(zip(errs, got)) if e or g[1] or g[0] is None]
    if bad:
        st["why"] = "; ".join(bad)
        return st

    st.update(ok=True, why="", set_flag=set_flag, wait_flag=wait_flag)
    return st


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps (default 100 = 0.35 s: behind the fence that opens the timed region the frame pipeline starts from "
                         "an idle GPU with its launches bunched and needs about six frames to spread them out, and the last launches "
                         "drain with nothing behind them -- together ~3 ms, profiles/r04_trace_frames.txt: 4 %% of a 20-step figure, "
                         "2 %% of a 40-step one, under 1 %% of this one; steady state is 3.41 ms per frame)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--spp", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--pipeline", type=int, default=0,
                    help="frame pipeline depth (rbrt_hip_scene_set_pipeline): consecutive steps' trace launches overlap "
                         "on this many internal streams; 0 = the library's automatic choice (8)")
    ap.add_argument("--vary-seed", type=int, default=0, help="1: every step renders a new frame (seed + step number)")
    ap.add_argument("--triangles", type=int, default=69451)
    ap.add_argument("--mesh", choices=("smooth", "rough"), default="smooth",
                    help="which stand-in takes bunny.obj's place: the smooth blob (headline) or the rough one (uneven triangle "
                         "sizes, concavities, thin parts; rbrt_amd/standin.py) -- brackets the headline's sensitivity to geometry")
    ap.add_argument("--scene", default=str(ROOT / "scenes" / "example_scene.yaml"))
    ap.add_argument("--config", choices=("2", "2r", "4", "4v"), default="2",
                    help="BASELINE configuration: 2 = the headline (69,451-triangle stand-in); 2r = the same with the ROUGH stand-in; "
                         "4 = the 871,414-triangle stand-in at the example scene's scale 45, where the reference's |a| >= 1e-3 rule "
                         "(triangle.rs:198-200) rejects every triangle of so fine a mesh: the mesh is traversed and never hit; 4v = the "
                         "same mesh at scale 450 (translation 50, -18, -145: the placement of the committed golden windows), where it is "
                         "VISIBLE: the deep-BVH stress test (GPU-built tree, LDS stack overflow figures in roofline.lds_stack)")
    ap.add_argument("--mesh-scale", type=float, default=None, help="overrides the scene file's mesh scale")
    ap.add_argument("--mesh-translation", default=None, help="x,y,z: overrides the scene file's mesh translation")
    ap.add_argument("--cpu-col-stride", type=int, default=-1,
                    help="the CPU baseline renders every n-th image column (0 = skip the CPU baseline; 1 = the whole "
                         "frame; default: the largest power-of-two fraction of the columns estimated to fit --cpu-budget-s)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU baseline threads (0 = every core this process may use)")
    ap.add_argument("--isolated-steps", type=int, default=8,
                    help="steps of the second, unpipelined leg that times isolated trace launches for the roofline (N = 1 only; 0 = skip)")
    ap.add_argument("--single-frames", type=int, default=5, help="blocking frames incl. D2H timed for single_frame (N = 1 only; 0 = skip)")
    ap.add_argument("--one-shot", type=int, default=5,
                    help="calls of the one-shot rbrt_hip_render (host arrays in, host RGB8 out: create + BVH build + render + copy + "
                         "destroy, what the reference's one render_scene call costs end to end) timed for one_shot (N = 1 only; 0 = skip)")
    ap.add_argument("--same-camera-steps", type=int, default=-1,
                    help="steps of the leg that renders ONE camera over and over (the tile pass cached); -1 = as many as --steps, 0 = skip")
    ap.add_argument("--cpu-budget-s", type=float, default=120.0,
                    help="the CPU baseline renders the largest power-of-two fraction of the image's columns estimated to fit this many seconds")
    ap.add_argument("--init-timeout-s", type=float, default=180.0, help="N > 1: limit for the process group's rendezvous and for every collective")
    ap.add_argument("--check", action="store_true", help="also compare the sampled columns with the GPU image")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="WORLD",
                    help="single process: time only rank 0's share of a WORLD-GPU run (its tiles, no gather); a "
                         "scaling estimate for one-GPU boxes, not a benchmark result")
    ap.add_argument("--emulate-rank", type=int, default=0, metavar="R", help="with --emulate-rank-of WORLD: the rank whose share is timed (default 0)")
    ap.add_argument("--emulate-all", default="", metavar="WORLDS",
                    help="single process, e.g. '2,4,8': time EVERY rank's share of a WORLD-GPU run, one after the other on one scene (its "
                         "tiles, no gather), and print one JSON record with per-rank ms per step, their max and mean, and predicted_scaling = "
                         "the whole frame's ms / the slowest rank's -- the step of an N-GPU run is the MAX over its ranks; not a benchmark line")
    ap.add_argument("--gather", choices=("auto", "rccl", "ipc"), default="auto",
                    help="N > 1: how rank 0 gets the other ranks' tiles. rccl: dist.gather (copy KERNELS, which need wave slots "
                         "beside the persistent trace launches); ipc: rank 0 maps the peers' buffers (HIP IPC) and pulls them with "
                         "device copies behind interprocess events (copy engines, no wave slots); auto: a short probe of both, the "
                         "faster one is used (rccl when the IPC set-up fails on any rank)")
    ap.add_argument("--gather-probe-steps", type=int, default=8, help="--gather auto: steps per probe leg (untimed, before the warm-up)")
    ap.add_argument("--rehearse-single-gpu", action="store_true",
                    help="N > 1 ranks all on cuda:0 with a gloo gather through host memory: exercises the sharded "
                         "path on a one-GPU box; its numbers mean nothing")
    return ap.parse_args()


def launch_ranks(n):
    """One rank per GPU under torch.distributed.run, as a child process of a parent that stays off the GPU. Returns the
    child's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:  # a free port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    sys.stderr.write(f"bench.py: --gpus {n} without a launcher: starting {' '.join(cmd[1:8])} ... as a child process\n")
    sys.stderr.flush()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL and the IPC gather need it on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.config == "2r":
        args.mesh = "rough"
    elif args.config in ("4", "4v"):
        args.triangles = 871414
        if args.config == "4v":
            args.mesh_scale = 450.0 if args.mesh_scale is None else args.mesh_scale
            args.mesh_translation = args.mesh_translation or "50,-18,-145"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` by itself: this process has not touched the GPU (torch is not even imported) and never
            # will; it starts the N ranks as a CHILD under torch.distributed.run (the fan-out of lib.rs:84-86), passes their
            # output through and exits with their code. No exec of any kind.
            sys.exit(launch_ranks(args.gpus))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import rbrt_amd
    from rbrt_amd import abi, standin

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.rehearse_single_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tmo = datetime.timedelta(seconds=args.init_timeout_s)
        try:
            if args.rehearse_single_gpu:
                dist.init_process_group("gloo", timeout=tmo)
            else:
                # the trace launches are persistent and hold every wave slot; the gather's copy kernels are short and
                # on the critical path of every step: RCCL's stream gets high priority (the library's trace streams low)
                os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        except Exception as e:  # a bounded wait and a message, never a hang and never a restart of a process that has touched the GPU
            sys.stderr.write(f"bench.py rank {rank}/{world}: the process group did not come up within {args.init_timeout_s:.0f} s "
                             f"({type(e).__name__}: {e}); MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')}\n")
            sys.stderr.flush()
            os._exit(3)

    # ---- setup (untimed): stand-in asset, YAML through the C++ host, upload + BVH build ---------
    work = Path(tempfile.mkdtemp(prefix=f"rbrt_bench_r{rank}_"))
    real_asset = Path("bunny.obj").exists() and args.triangles == standin.BUNNY_TRIANGLES and args.mesh == "smooth"
    obj = Path("bunny.obj").resolve() if real_asset else standin.ensure_obj(work / "bunny.obj", args.triangles, args.mesh)
    yaml_text = Path(args.scene).read_text().replace("obj_filepath: bunny.obj", f"obj_filepath: {obj}")
    if args.mesh_scale is not None or args.mesh_translation:
        import yaml
        doc = yaml.safe_load(yaml_text)
        for mb in doc.get("mesh_blueprints") or []:
            if args.mesh_scale is not None:
                mb["scale"] = float(args.mesh_scale)
            if args.mesh_translation:
                mb["translation"] = dict(zip("xyz", (float(v) for v in args.mesh_translation.split(","))))
        yaml_text = "---\n" + yaml.safe_dump(doc, sort_keys=False)
    (work / "scene.yaml").write_text(yaml_text)
    devnull = os.open(os.devnull, os.O_WRONLY)  # the host prints the reference's progress lines
    saved = os.dup(1)
    os.dup2(devnull, 1)
    try:
        host_scene = abi.HostScene(work / "scene.yaml", args.height, args.width)
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
    cam = host_scene.camera
    W, H, spp = args.width, args.height, args.spp
    t0 = time.perf_counter()
    scene = rbrt_amd.HipScene(host_scene, device=local_rank)
    setup_s = time.perf_counter() - t0
    # The handle starts with the tree that costs scene_create least (the device builder's) and adopts the host builder's
    # tree when a background thread has made it (api.cpp struct Refine): the counting pass and every timed leg below run on
    # the tree a handle that goes on rendering ends up with, so it is waited for here. The one_shot leg pays for its own.
    create_times = scene.create_times()
    refine_state, refine_s = scene.refine_wait(300.0)
    setup_total_s = time.perf_counter() - t0
    emu = args.emulate_rank_of if world == 1 and args.emulate_rank_of > 1 else 0
    if emu and not 0 <= args.emulate_rank < emu:
        raise SystemExit("--emulate-rank must be a rank of --emulate-rank-of")
    opts = abi.default_opts(spp=spp, seed=args.seed, tile_rank=args.emulate_rank if emu else rank, tile_world=emu if emu else world)
    if os.environ.get("RBRT_BENCH_MAX_DEPTH"):  # diagnosis only (changes the image)
        opts.max_depth = int(os.environ["RBRT_BENCH_MAX_DEPTH"])
    stream = torch.cuda.current_stream().cuda_stream

    import numpy as np_mod

    def guarded_early(fn):
        try:
            return fn()
        except Exception as e:
            sys.stderr.write(f"bench.py rank {rank}/{world}: a collective failed during set-up ({type(e).__name__}: {e}); no result line\n")
            sys.stderr.flush()
            os._exit(4)

    image = torch.empty((H, W, 3), dtype=torch.float32, device=dev) if rank == 0 else None
    if world > 1:
        slot_pixels = rbrt_amd.packed_pixels(W, H, 0, world)  # rank 0 owns the most tiles: equal-size slots
        maxn = slot_pixels * 3
        # NBUF sets of buffers and a side stream: the gather of frame k (and rank 0's unpack) runs beside the renders of
        # the frames after it instead of in front of them -- the render's resolve kernel, which frees a pipeline lane for
        # its next launch, does not queue behind a collective that waits for the slowest rank. Four sets (round 4; two
        # before): the collective's kernels are multi-wave workgroups and, like every kernel beside the persistent trace
        # launches, start only when a launch drains (DESIGN.md section 6 "Short kernels beside a persistent one"); a step
        # of an eighth is 0.5 ms, so a gather may take three steps before a render has to wait for its buffer.
        NBUF = 4
        mine = [torch.empty(maxn, dtype=torch.float32, device=dev) for _ in range(NBUF)]  # this rank's tiles (tail of the slot unused)
        # rank 0 receives straight into one buffer of `world` equal slots; the unpack kernel strides over it
        slots = [torch.empty(world * maxn, dtype=torch.float32, device=dev) for _ in range(NBUF)] if rank == 0 else None
        gathered = [list(sl.chunk(world)) for sl in slots] if rank == 0 else [None] * NBUF
        xs = torch.cuda.Stream(device=dev, priority=-1)  # (high priority: short kernels among persistent ones)
        rendered = [torch.cuda.Event() for _ in range(NBUF)]
        gathered_ev = [None] * NBUF
        # The IPC gather (--gather ipc / auto). Every rank's NBUF buffers and an interprocess event per buffer are opened by
        # rank 0 (torch's CUDA-IPC reductions: hipIpcGetMemHandle / hipIpcOpenMemHandle, hipIpcGetEventHandle); rank 0 waits
        # for a peer's event, copies the peer's buffer into its slot with a device copy (xGMI, the copy engines), and records
        # an interprocess event of its own per buffer that the peers wait for before they render into that buffer again.
        # A wait on an interprocess event takes the LAST record issued, so the record has to have been issued (not run)
        # before the wait is: each rank publishes the number of the step it has issued in a small file in /dev/shm, and a
        # rank checks that number on the host before it enqueues a wait (host threads run ahead of the GPUs; nobody waits
        # for a GPU there).
        ipc = {"ok": False, "why": "not asked for (--gather rccl)"}
        if args.gather in ("auto", "ipc"):
            ipc = setup_ipc_gather(torch, dist, np_mod, rank, world, dev, mine, NBUF, guarded_early, timeout_s=args.init_timeout_s)
            if args.gather == "ipc" and not ipc["ok"]:
                sys.stderr.write(f"bench.py rank {rank}/{world}: --gather ipc cannot be set up ({ipc['why']})\n")
                os._exit(5)
        gather_mode = ["ipc" if args.gather == "ipc" else "rccl"]  # (auto: decided by the probe below)

    step_no = [0]
    cam_no = [0]
    import numpy as np

    cam_x = [np.float32(cam.position[0])]

    def fresh_camera():
        """A camera the library has not seen: the position's first component moves one unit in the last place per call
        (every rank makes the same sequence). The image and the work are the frame's to within rounding; the library's
        per-camera tile pass has to run again. (Constant time per call: an earlier form walked from the first camera to the
        k-th on every call, and after a thousand frames the HOST took longer per step than an eighth of a frame does.)"""
        cam_no[0] += 1
        c = type(cam).from_buffer_copy(cam)
        cam_x[0] = np.nextafter(cam_x[0], np.float32(np.inf), dtype=np.float32)
        c.position[0] = float(cam_x[0])
        return c

    if args.pipeline > 0:  # 0: leave the library's choice (automatic, or $RBRT_PIPELINE)
        scene.set_pipeline(args.pipeline)

    phase_ev = []  # N > 1: (before render, rendered, gather begins, gathered, unpacked) per timed step, for the per-rank split

    def step(new_camera=False, timed=False, final=False):
        step_no[0] += 1
        if args.vary_seed:  # (tests: a stale buffer shows; the frame whose hash is reported has the configuration's seed)
            opts.seed = args.seed if final else args.seed + step_no[0]
        c = fresh_camera() if new_camera else cam
        if world == 1:
            scene.render_device(c, opts, image.data_ptr(), None, stream)  # (emulation: packed tiles, fits)
            return
        b = step_no[0] % NBUF
        k = step_no[0]
        use_ipc = gather_mode[0] == "ipc"
        main = torch.cuda.current_stream()
        if gathered_ev[b] is not None:
            main.wait_event(gathered_ev[b])  # the gather of NBUF frames ago has read this buffer
        if rank != 0 and ipc["ok"] and ipc["last_use"][b] > 0:
            # (rank 0's copy of what this buffer held, whichever way THIS step gathers: its record has been issued once rank 0
            # says it has consumed that step)
            ipc["wait_flag"](world, ipc["last_use"][b])
            main.wait_event(ipc["consumed_by_rank0"][b])
            ipc["last_use"][b] = 0
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if timed else None
        if evs:
            evs[0].record(main)
        # (IPC gather: rank 0 renders straight into its slot of the gathered buffer -- a copy inside one device is a blit
        # KERNEL, which would wait for wave slots like every multi-wave kernel beside the persistent launches)
        out_buf = gathered[b][0] if use_ipc and rank == 0 else mine[b]
        scene.render_device(c, opts, out_buf.data_ptr(), None, stream)
        if evs:
            evs[1].record(main)  # (ahead of the event the side stream waits for: the spans below cannot come out negative)
        rendered[b].record(main)
        if use_ipc:
            ipc["rendered"][b].record(main)
            ipc["last_use"][b] = k
            ipc["set_flag"](rank, k)
        xs.wait_event(rendered[b])
        with torch.cuda.stream(xs):
            if evs:
                evs[2].record(xs)
            if use_ipc:
                if rank == 0:
                    for r in range(1, world):
                        ipc["wait_flag"](r, k)  # (host: rank r has issued its record for this step)
                        xs.wait_event(ipc["peer_rendered"][r][b])
                        gathered[b][r].copy_(ipc["peer_mine"][r][b], non_blocking=True)
            elif args.rehearse_single_gpu:  # gloo cannot gather device tensors: stage through the host
                host = mine[b].cpu()
                hg = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, hg, dst=0)
                if rank == 0:
                    for r in range(world):
                        gathered[b][r].copy_(hg[r], non_blocking=False)
            else:
                dist.gather(mine[b], gathered[b], dst=0)  # RCCL over xGMI: every peer sends its tiles straight to rank 0
            if evs:
                evs[3].record(xs)
            if rank == 0:
                rbrt_amd.unpack_tiles(local_rank, slots[b].data_ptr(), W, H, world, image.data_ptr(), None, xs.cuda_stream,
                                      rank_stride_pixels=slot_pixels)
                if use_ipc:  # (the peers' buffers of this step have been read once the copies above have run)
                    ipc["consumed"][b].record(xs)
                    ipc["set_flag"](world, k)
            ev = torch.cuda.Event()
            ev.record(xs)
            gathered_ev[b] = ev
            if evs:
                evs[4].record(xs)
                phase_ev.append(evs)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- counting pass (untimed): the work counters behind the algorithmic-bytes figure ----------
    stats_opts = abi.default_opts(spp=spp, seed=args.seed, tile_rank=args.emulate_rank if emu else rank, tile_world=emu if emu else world,
                                  flags=abi.FLAG_COLLECT_STATS)
    n_out = rbrt_amd.packed_pixels(W, H, rank, world) * 3 if world > 1 else W * H * 3
    scratch = torch.empty(n_out, dtype=torch.float32, device=dev)
    scene.render_device(cam, stats_opts, scratch.data_ptr(), None, stream)
    torch.cuda.synchronize()
    st = scene.stats()
    dbg_all = scene.debug_counters()
    sinfo = scene.info()
    dbg = dbg_all if os.environ.get("RBRT_BENCH_DEBUG") else None
    del scratch

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.rehearse_single_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def guarded(fn):
        """N > 1: a failed collective ends the run with a message and a non-zero exit code (the watchdog's timeout bounds the
        wait); the process is never restarted or re-exec'ed once it has touched the GPU."""
        try:
            return fn()
        except Exception as e:
            if world == 1:
                raise
            sys.stderr.write(f"bench.py rank {rank}/{world}: a collective failed ({type(e).__name__}: {e}); no result line\n")
            sys.stderr.flush()
            os._exit(4)

    # ---- the timed region: K steps, every one with a camera the library has not seen (the tile pass runs every step) ----
    helpers_issued = [0]  # helper launches of the last timed leg (the library's elastic launches: more waves for launches already running)

    def timed_leg(n_warm, n_steps, new_camera):
        for _ in range(n_warm):
            step(new_camera)
        fence()
        scene.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(new_camera, timed=world > 1 and new_camera)
        enq = time.perf_counter() - t0  # host time to issue the steps (they run asynchronously)
        fence()
        el = max_over_ranks(time.perf_counter() - t0)
        tr, rs, nl = scene.kernel_ms()
        mix = scene.launch_mix()  # the grid of a launch depends on what was in flight when it was issued
        helpers_issued[0] = scene.helper_launches()
        scene.set_timing(False)
        return el, enq, tr, rs, nl, mix

    # ---- set-up (untimed): the library makes the frame pipeline's lanes and their buffers when it first sees a STREAM of
    # calls (one issued while another is running): three calls back to back here, whatever --warmup is ----
    def prime_pipeline():
        for _ in range(3):
            step(True)
        fence()
    guarded(prime_pipeline)
    if world == 1 and args.emulate_all:
        # Every rank's share, not rank 0's: the step of an N-GPU run is the MAX over its ranks (the reference balances by work
        # stealing, lib.rs:84-88; here a rank's tiles lie on skew lines through the image, rbrt_hip.h "How tiles are dealt to ranks").
        worlds = [int(x) for x in args.emulate_all.split(",") if x.strip()]
        table = {}
        whole = None
        for wd in [1] + [w_ for w_ in worlds if w_ > 1]:
            # (two sweeps over the ranks, first to last and last to first, averaged: a GPU under a minute of steady load drifts by
            # several per cent, which a single sweep would book to the ranks measured last)
            acc = [0.0] * wd
            for order in (range(wd), reversed(range(wd))):
                for r in order:
                    opts.tile_rank, opts.tile_world = r, wd
                    acc[r] += timed_leg(args.warmup, args.steps, True)[0] / args.steps * 1e3
            per_rank = [round(a / 2.0, 4) for a in acc]
            if wd == 1:
                whole = per_rank[0]
                continue
            table[str(wd)] = {"ms_per_rank": per_rank, "max_ms": max(per_rank), "mean_ms": round(sum(per_rank) / wd, 4),
                              "max_over_mean": round(max(per_rank) / (sum(per_rank) / wd), 4),
                              "predicted_scaling": round(whole / max(per_rank), 3),
                              "local_tiles": [rbrt_amd.packed_pixels(W, H, r, wd) // 64 for r in range(wd)]}
        opts.tile_rank, opts.tile_world = 0, 1
        scene.check()
        print(json.dumps({"what": "every rank's share of an N-GPU run timed on ONE GPU, one rank after the other (no gather): ms per step of "
                                  f"{args.steps} pipelined steps, every step a new camera; predicted_scaling = whole frame / slowest rank",
                          "workload": f"{args.triangles}-triangle {'stand-in' if args.mesh == 'smooth' else 'ROUGH stand-in'}, {W}x{H}, {spp} spp, config {args.config}",
                          "whole_frame_ms": whole, "worlds": table, "steps": args.steps, "warmup": args.warmup}), flush=True)
        return
    # ---- N > 1, --gather auto: both gathers for a few untimed steps each, the faster one is used from here on ----
    gather_probe = None
    if world > 1 and args.gather == "auto" and ipc["ok"]:
        def probe():
            times = {}
            for mode in ("rccl", "ipc"):
                gather_mode[0] = mode
                for _ in range(2):
                    step(True)
                fence()
                t0 = time.perf_counter()
                for _ in range(args.gather_probe_steps):
                    step(True)
                fence()
                times[mode] = max_over_ranks(time.perf_counter() - t0) / args.gather_probe_steps * 1e3
            return times
        gather_probe = guarded(probe)
        gather_mode[0] = "ipc" if gather_probe["ipc"] < gather_probe["rccl"] else "rccl"
    elapsed, enqueue_s, trace_ms, resolve_ms, n_launches, (mix_full, mix_half) = guarded(lambda: timed_leg(args.warmup, args.steps, True))
    helpers_timed = helpers_issued[0]
    scene.check()  # a NaN sphere discriminant (sphere.rs:33 panics) or corrupt path state fails the run loudly
    # per-rank split of a step (N > 1): spans on this rank's streams, means over the timed steps
    phases = None
    if world > 1:
        def mean_ms(a, b):
            return sum(e[a].elapsed_time(e[b]) for e in phase_ev) / max(1, len(phase_ev))
        mine_ph = {"rank": rank, "render_ms": round(mean_ms(0, 1), 4), "gather_wait_ms": round(mean_ms(1, 2), 4),
                   "gather_ms": round(mean_ms(2, 3), 4), "unpack_ms": round(mean_ms(3, 4), 4)}
        allp = [None] * world
        guarded(lambda: dist.all_gather_object(allp, mine_ph))
        phases = {"per_rank": allp,
                  "max": {k: max(p[k] for p in allp) for k in ("render_ms", "gather_wait_ms", "gather_ms", "unpack_ms")},
                  "what": "spans between events on each rank's own streams, mean over the timed steps: render = the library's call on the "
                          "caller's stream (trace launches on its lanes + resolve); gather_wait = side stream waiting for it; gather = "
                          "dist.gather, or rank 0's copies out of the peers' buffers (collective.gather says which); unpack = rank 0's de-interleave"}
    # ---- the same camera over and over (a stream of frames of one view: the tile pass is paid once) ----
    same = None
    n_same = args.steps if args.same_camera_steps < 0 else args.same_camera_steps
    if n_same > 0:
        s_el, _, s_tr, _, s_nl, s_mix = guarded(lambda: timed_leg(args.warmup, n_same, False))
        same = {"steps": n_same, "ms_per_step": s_el / n_same * 1e3, "kernel_ms_pipelined": s_tr / max(1, s_nl),
                "launch_mix": {"full_grid": s_mix[0], "half_grid": s_mix[1]}}

    # ---- second leg (N = 1): isolated launches. With the pipeline on, a launch's event span includes time it shares
    # with its neighbours (it is a latency, and can exceed ms_per_step); the roofline's denominator is the
    # duration of a launch that has the GPU to itself, measured here with HIP events on the launch's own stream.
    iso = None
    if world == 1 and not emu and args.isolated_steps > 0 and args.pipeline != 1:
        scene.set_pipeline(1)
        step()
        fence()
        scene.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(args.isolated_steps):
            step()
        fence()
        iso_elapsed = time.perf_counter() - t0
        iso_trace_ms, iso_resolve_ms, iso_n = scene.kernel_ms()
        scene.set_timing(False)
        scene.set_pipeline(args.pipeline)
        iso = {"steps": args.isolated_steps, "ms_per_step": iso_elapsed / args.isolated_steps * 1e3,
               "kernel_ms": iso_trace_ms / max(1, iso_n), "resolve_ms": iso_resolve_ms / max(1, iso_n), "launches": iso_n}
    # ---- third leg (N = 1): one blocking frame at a time, image copied to host memory (pinned) --------------------
    # Two forms: what the reference's render_scene RETURNS is the 8-bit RGB image (lib.rs:75-79, :116-122): the frame is
    # rendered with the quantisation done on the device and 2.4 MB go over PCIe; and, as in rounds 1-2, the fp32 radiance
    # (9.4 MB), which the reference never exposes.
    single = None
    if world == 1 and not emu and args.single_frames > 0:
        host_img = torch.empty((H, W, 3), dtype=torch.float32).pin_memory()
        host_rgb = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
        rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        step()  # (untimed: the first blocking frame after a stream of frames is still issued as one of the stream)
        fence()
        ts, ts8 = [], []
        for _ in range(args.single_frames):
            t0 = time.perf_counter()
            step()
            host_img.copy_(image, non_blocking=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        for _ in range(args.single_frames):
            t0 = time.perf_counter()
            scene.render_device(cam, opts, None, rgb8.data_ptr(), stream)
            host_rgb.copy_(rgb8, non_blocking=True)
            torch.cuda.synchronize()
            ts8.append(time.perf_counter() - t0)
        tsn = []
        for _ in range(args.single_frames):  # ... of a camera the library has not seen: with the tile pass
            c = fresh_camera()
            t0 = time.perf_counter()
            scene.render_device(c, opts, None, rgb8.data_ptr(), stream)
            host_rgb.copy_(rgb8, non_blocking=True)
            torch.cuda.synchronize()
            tsn.append(time.perf_counter() - t0)
        single = {"frames": args.single_frames, "ms": sum(tsn) / len(tsn) * 1e3, "ms_min": min(tsn) * 1e3,
                  "ms_same_camera": sum(ts8) / len(ts8) * 1e3, "ms_radiance": sum(ts) / len(ts) * 1e3}

    # ---- fourth leg (N = 1): the ONE call the reference makes (src/main.rs:82), end to end: rbrt_hip_render from the host's
    # scene arrays to the host's RGB8 image -- scene upload, BVH build, lanes, tile pass, render, quantisation, copy,
    # release -- of a camera the library has not seen. Nothing of it is in `value`.
    one_shot = None
    if world == 1 and not emu and args.one_shot > 0:
        torch.cuda.synchronize()
        rad1, _ = rbrt_amd.render_scene(cam, spp, host_scene, seed=args.seed)  # (untimed: the configuration's camera, for the hash)
        first = rbrt_amd.last_render_times()
        recs = []
        for _ in range(args.one_shot):
            c = fresh_camera()
            t0 = time.perf_counter()
            rbrt_amd.render_scene(c, spp, host_scene, seed=args.seed, want_radiance=False)
            recs.append((time.perf_counter() - t0, rbrt_amd.last_render_times()))
        # (the parts are those of the MEDIAN call: a box shared with other tenants now and then takes three times as long
        # over the call's two dozen host round trips -- allocations, synchronisations --, and a mean of five is then neither)
        import statistics
        recs.sort(key=lambda r: r[0])
        med = recs[len(recs) // 2]
        mean = lambda k: med[1][k] * 1e3  # noqa: E731
        import hashlib as _h
        one_shot = {"calls": len(recs), "ms": round(statistics.median(r[0] for r in recs) * 1e3, 3), "ms_min": round(recs[0][0] * 1e3, 3),
                    "ms_max": round(recs[-1][0] * 1e3, 3), "ms_mean": round(sum(r[0] for r in recs) / len(recs) * 1e3, 3),
                    "create_ms": round(mean("create_s"), 3), "upload_ms": round(mean("upload_s"), 3), "build_ms": round(mean("bvh_build_s"), 3),
                    "lanes_ms": round(mean("lanes_s"), 3), "render_ms": round(mean("render_s"), 3), "copy_ms": round(mean("copy_s"), 3),
                    "destroy_ms": round(mean("destroy_s"), 3),
                    "bvh_builder": "device" if recs[-1][1]["meshes_device_built"] else "host",
                    "first_call_ms": round(first["total_s"] * 1e3, 3),
                    "image_sha256_16": _h.sha256(rad1.tobytes()).hexdigest()[:16],
                    "value": round(W * H * spp / statistics.median(r[0] for r in recs) / 1e6, 2), "unit": "Mray-samples/s",
                    "what": "rbrt_hip_render, the drop-in for the reference's one render_scene call (src/main.rs:82), from the host's scene "
                            "arrays to the host's RGB8 image, a camera the library has not seen: create (upload + BVH build by whichever "
                            "builder costs the call less + lanes) + render (tile pass, trace, resolve, quantise) + copy + destroy; ms = the median "
                            "call, whose parts the *_ms fields are; the HIP runtime is up already (first_call_ms: this process's first such call, with the radiance copied too)"}
        del rad1

    # the frame whose hash is reported: the configuration's own camera, rendered last (untimed)
    guarded(lambda: (step(False, final=True), fence()))
    if world > 1 and ipc["ok"]:  # rank 0 lets go of the peers' buffers before the peers (who end here) free them
        import gc
        for key in ("peer_mine", "peer_rendered", "consumed_by_rank0"):
            ipc.pop(key, None)
        gc.collect()
        torch.cuda.ipc_collect()
        guarded(lambda: dist.barrier())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    import hashlib
    image_sha = hashlib.sha256(image.cpu().numpy().tobytes()).hexdigest()[:16]
    if os.environ.get("RBRT_PRIMARY_CULL") == "0" and os.environ.get("RBRT_HIP_LAB") == "1":
        tile_pass = "off (lab knob RBRT_PRIMARY_CULL=0)"
    else:
        table = scene.primary_cull(cam)  # (debug hook: the table the library computes for this camera)
        tile_pass = {"tiles": int(table.size), "background_only_tiles": int((table >> 31).sum()),
                     "finished_by": "sky_resolve_kernel (streaming), the rest by trace_megakernel + resolve_kernel",
                     # every timed step renders a camera the library has not seen: primary_cull_kernel + tile_lists_kernel
                     # run inside the timed region, every step (ms_per_step_same_camera: the leg where they are cached)
                     "in_timed_region": True, "cameras_in_timed_region": args.steps}
    samples_per_step = W * H * spp
    if emu:
        samples_per_step = rbrt_amd.packed_pixels(W, H, 0, emu) * spp
    value = samples_per_step * args.steps / elapsed / 1e6
    # dominant kernel = trace_kernel; algorithmic bytes of ONE launch on this rank (DESIGN.md "Measurement")
    local_pixels = (rbrt_amd.packed_pixels(W, H, rank, world) if world > 1 else W * H)
    alg_bytes = (st["nodes_visited"] * st["node_bytes"] + st["tris_tested"] * TRI_ALG_BYTES +
                 st["mesh_hits"] * NORMAL_ALG_BYTES + local_pixels * PIXEL_ALG_BYTES)
    pipelined_kernel_ms = trace_ms / max(1, n_launches)
    launches_per_step = max(1, round(n_launches / max(1, args.steps)))  # > 1 when a frame needs several sample batches
    alg_bytes = alg_bytes // launches_per_step                          # the counters cover the whole frame
    if iso is not None:
        kernel_ms, kernel_ms_from = iso["kernel_ms"], "isolated launches (second leg, pipeline 1)"
    else:
        kernel_ms = pipelined_kernel_ms
        kernel_ms_from = ("the timed region (pipeline 1: launches do not overlap)" if args.pipeline == 1 else
                          "the timed region, launches OVERLAP: a latency, not the cost of the work")
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "numerator": "ALGORITHMIC bytes (SURVEY 8(d)): 128 B x BVH nodes visited + 36 B x triangles tested + 12 B x "
                             "mesh hits + 12 B x pixels; mostly L1/L2 hits, not DRAM traffic",
                "kernel": "trace_megakernel", "kernel_ms": round(kernel_ms, 4), "kernel_ms_from": kernel_ms_from,
                "kernel_ms_pipelined": round(pipelined_kernel_ms, 4), "launches_timed": n_launches,
                "algorithmic_bytes_per_launch": alg_bytes,
                "counters": {k: st[k] for k in ("rays", "mesh_gate_pass", "nodes_visited", "tris_tested", "mesh_hits")},
                "per_sample": {"rays": round(st["rays"] / max(1, st["samples"]), 3), "nodes_visited": round(st["nodes_visited"] / max(1, st["samples"]), 3),
                               "tris_tested": round(st["tris_tested"] / max(1, st["samples"]), 3),
                               "nodes_per_traversal": round(st["nodes_visited"] / max(1, st["mesh_gate_pass"]), 2)},
                # the per-lane traversal stack: kLdsStack entries in LDS, deeper ones exactly in a per-wave global scratch
                "lds_stack": {"entries_in_lds": int(os.environ.get("RBRT_LDS_STACK", "8")) if os.environ.get("RBRT_HIP_LAB") == "1" else 8,
                              "tree_stack_need": sinfo.get("bvh_stack_need"),
                              "pushes_beyond_lds": dbg_all["stack_pushes_beyond_lds"],
                              "pushes_beyond_lds_per_traversal": round(dbg_all["stack_pushes_beyond_lds"] / max(1, st["mesh_gate_pass"]), 5),
                              "deepest_stack": dbg_all["stack_deepest"],
                              "bvh_builder": ("host" if not create_times["meshes_device_built"] else
                                              f"device first ({create_times['bvh_build_s'] * 1e3:.1f} ms), the host builder's tree adopted from a "
                                              f"background thread after {refine_s * 1e3:.0f} ms" if refine_state == 1 else "device"),
                              "bvh_nodes": sinfo.get("n_nodes")}}
    if iso is not None:
        roofline["isolated_leg"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in iso.items()}
    # The two figures that describe the kernel as it really runs come from rocprofv3 PMC passes (their own runs of
    # this same command, tools/profile.sh; committed under profiles/): a bench run cannot read counters itself. A
    # summary is used only if it was measured on the kernel sources this run uses (rbrt_amd/srchash.py); otherwise the
    # counter-derived figures are left out and the line says so.
    if world == 1 and not emu:
        from rbrt_amd.srchash import kernel_source_sha256
        src_hash = kernel_source_sha256()
        roofline["kernel_source_sha256_16"] = src_hash
        cands = sorted((ROOT / "profiles").glob("r*_pmc_summary.json"), reverse=True)
        rec, prof = None, None
        for cand in cands:
            try:
                r = json.loads(cand.read_text())
            except Exception:
                continue
            if (r.get("workload") == f"{W}x{H}x{spp}" and r.get("triangles") == args.triangles and args.mesh == "smooth" and
                    r.get("kernel_source_sha256_16") == src_hash):
                rec, prof = r, cand
                break
        if rec is None:
            roofline["counters_stale"] = True
            roofline["counters_note"] = ("no profiles/r*_pmc_summary.json was measured on these kernel sources "
                                         f"(sha {src_hash}) and workload: traffic / measured_hbm / valu_issue omitted; run tools/profile.sh")
        else:
            src = f"profiles/{prof.name}"
            roofline["counters_stale"] = False
            roofline["traffic"] = rec.get("hbm_bytes_per_launch")
            roofline["traffic_source"] = f"{src} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
            if rec.get("hbm_bytes_per_launch") and kernel_ms > 0:
                gbs = rec["hbm_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
                roofline["measured_hbm"] = {"bytes_per_launch": rec["hbm_bytes_per_launch"], "GBps": round(gbs, 1),
                                            "frac": round(gbs / HBM_PEAK_GBS, 4)}
            if rec.get("sq_insts_valu") and kernel_ms > 0:
                # SIMD cycles per wave64 VALU instruction: measured, tools/ubench/valu_rate.hip -> profiles/r03_valu_rate.txt
                # (plain f32 / integer VALU at four or more resident waves per SIMD; one wave alone issues one per
                # VALU_CYCLES_ONE_WAVE cycles -- the figure rounds 1 and 2 priced the kernel with, which is a LATENCY).
                bound_ms = rec["sq_insts_valu"] * VALU_CYCLES_PER_INST / 1024.0 / CLOCK_GHZ / 1e9 * 1e3
                roofline["valu_issue"] = {"sq_insts_valu": rec["sq_insts_valu"], "cycles_per_inst": VALU_CYCLES_PER_INST,
                                          "cycles_per_inst_source": "profiles/r03_valu_rate.txt (tools/ubench/valu_rate.hip)",
                                          "simds": 1024, "clock_ghz": CLOCK_GHZ, "bound_ms": round(bound_ms, 3),
                                          "frac": round(bound_ms / kernel_ms, 4),
                                          "one_wave_cycles_per_inst": VALU_CYCLES_ONE_WAVE,
                                          "valu_lane_utilisation": rec.get("valu_lane_utilisation"), "source": src}
                if rec.get("sq_insts_salu") and rec.get("sq_insts_lds") is not None:
                    valu_mean, mix_from = ISSUE_COST["valu_mean"], "static mix of the binary (tools/static_cost.py): 62 % full-rate, 35.5 % other, 2.5 % transcendental"
                    vm = rec.get("valu_mix")
                    if vm and rec["sq_insts_valu"]:
                        # the EXECUTED mix (rocprofv3 SQ_INSTS_VALU_* in the same summary): f32 add / mul / fma issue at the full
                        # rate, transcendentals at 6.1 cycles, the integer classes at the full rate for INT_FULL_FRAC of them (the
                        # share of add / xor / and / or / shift among the binary's integer instructions), everything else 3.1
                        full = vm.get("add_f32", 0) + vm.get("mul_f32", 0) + vm.get("fma_f32", 0) + INT_FULL_FRAC * vm.get("int32", 0)
                        trans = vm.get("trans_f32", 0)
                        other = rec["sq_insts_valu"] - full - trans
                        valu_mean = (full * 2.0 + other * 3.1 + trans * 6.1) / rec["sq_insts_valu"]
                        mix_from = (f"executed mix (SQ_INSTS_VALU_*): {full / rec['sq_insts_valu']:.3f} full-rate, {other / rec['sq_insts_valu']:.3f} other, "
                                    f"{trans / rec['sq_insts_valu']:.3f} transcendental")
                    cyc = (rec["sq_insts_valu"] * valu_mean + rec["sq_insts_salu"] * ISSUE_COST["salu"] +
                           rec.get("sq_insts_lds", 0) * ISSUE_COST["lds"] + rec.get("sq_insts_vmem_rd", 0) * ISSUE_COST["vmem"])
                    mix_ms = cyc / 1024.0 / CLOCK_GHZ / 1e9 * 1e3
                    roofline["issue_mix"] = {"bound_ms": round(mix_ms, 3), "frac": round(mix_ms / kernel_ms, 4),
                                             "cycles_per_class": {**{k: round(v, 3) for k, v in ISSUE_COST.items()}, "valu_mean": round(valu_mean, 3)},
                                             "valu_mix_from": mix_from,
                                             "insts": {"valu": rec["sq_insts_valu"], "salu": rec["sq_insts_salu"], "lds": rec.get("sq_insts_lds"),
                                                       "vmem_rd": rec.get("sq_insts_vmem_rd")},
                                             "what": "SIMD issue cycles of ALL instruction classes at their measured costs (profiles/r03_valu_rate.txt) "
                                                     "/ 1024 SIMDs / 2.4 GHz: the bound that binds (the tree is served from L1/L2); a model "
                                                     "good to about one digit -- read it as ~0.8, not as three figures"}

    out = {
        "metric": "Mray-samples/sec (WxHxspp/s) on bunny scene; achieved HBM GB/s vs peak",
        "value": round(value, 2), "unit": "Mray-samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "ms_per_step_new_camera": round(elapsed / args.steps * 1e3, 4),  # (= ms_per_step: every timed step has a new camera)
        "ms_per_step_same_camera": round(same["ms_per_step"], 4) if same else None,
        "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"example_scene.yaml, {args.triangles}-triangle "
                               f"{'bunny.obj' if real_asset else 'stand-in mesh' if args.mesh == 'smooth' else 'ROUGH stand-in mesh'}"
                               f"{'' if args.mesh_scale is None else f' at scale {args.mesh_scale:g} (translation {args.mesh_translation})'}, "
                               f"{W}x{H}, {spp} spp, seed {args.seed}",
                   "baseline_config": args.config,
                   "parallelism": f"pixel tiles 8x8 dealt round-robin along skew lines over {world} GPU(s)" + (f", {gather_mode[0]} gather to rank 0" if world > 1 else ""),
                   "pipeline": pipeline_note(args.pipeline),
                   "host_issue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4),
                   "launch_mix_timed_region": {"full_grid": mix_full, "half_grid": mix_half},
                   # (the library's elastic launches: when the caller stops issuing and the launches in flight leave wave slots
                   # free -- the end of the timed region's stream --, a watcher thread gives them more waves: helper launches)
                   "helper_launches_timed_region": helpers_timed,
                   # which tiles the trace kernel never sees (DESIGN.md "The tile pass"): every sample of theirs is still
                   # produced -- by sky_resolve_kernel, inside the timed region -- and counted in `value`
                   "tile_pass": tile_pass,
                   "setup_s_excluded": round(setup_total_s, 3),
                   "setup": {"scene_create_s": round(setup_s, 4), "hip_init_s": round(create_times["hip_init_s"], 4),
                             "upload_s": round(create_times["upload_s"], 4), "bvh_build_s": round(create_times["bvh_build_s"], 4),
                             "lanes_s": round(create_times["lanes_s"], 4), "waited_for_the_background_tree_s": round(setup_total_s - setup_s, 4),
                             "background_tree": {0: "none started", 1: "adopted", 2: "failed or cancelled", 3: "still at work"}[refine_state]},
                   "image_sha256_16": image_sha,
                   **({"EMULATION_rank0_share_of_world": emu} if emu else {}),
                   # (with the pipeline on, the resolve waits on another stream: its event pair measures that wait)
                   "resolve_kernel_ms": round(resolve_ms / max(1, n_launches), 4) if args.pipeline == 1 else None},
        "roofline": roofline,
    }
    if same is not None:
        out["same_camera_leg"] = {"steps": same["steps"], "ms_per_step": round(same["ms_per_step"], 4),
                                  "value": round(samples_per_step / (same["ms_per_step"] * 1e-3) / 1e6, 2),
                                  "kernel_ms_pipelined": round(same["kernel_ms_pipelined"], 4), "launch_mix": same["launch_mix"],
                                  "what": "the same frame rendered over and over: the library keeps the tile pass's lists while camera and "
                                          "tile partition stay what they were (a fixed view, e.g. progressive refinement); NOT `value`"}
    if phases is not None:
        out["phases"] = phases
        out["collective"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                             "library": ("gloo (rehearsal on one GPU through host memory)" if args.rehearse_single_gpu else
                                         "RCCL " + ".".join(str(x) for x in torch.cuda.nccl.version())),
                             "gather_bytes_per_rank": int(maxn * 4), "timeout_s": args.init_timeout_s,
                             "gather": {"rccl": "dist.gather (" + dist.get_backend() + ")",
                                        "ipc": "rank 0 copies out of the peers' buffers (HIP IPC, interprocess events)"}[gather_mode[0]],
                             "gather_asked": args.gather, "gather_probe_ms_per_step": gather_probe,
                             "ipc_gather_available": bool(ipc["ok"]), "ipc_gather_note": ipc["why"] or None}
    if one_shot is not None:
        out["one_shot"] = one_shot
    if single is not None:
        out["single_frame"] = {"ms": round(single["ms"], 4), "ms_new_camera": round(single["ms"], 4), "ms_min": round(single["ms_min"], 4),
                               "ms_same_camera": round(single["ms_same_camera"], 4), "frames": single["frames"],
                               "value": round(samples_per_step / (single["ms"] * 1e-3) / 1e6, 2), "unit": "Mray-samples/s",
                               "ms_with_fp32_radiance_copy": round(single["ms_radiance"], 4),
                               "what": "one blocking frame of a camera the library has not seen, as the reference's render_scene returns "
                                       "it: tile pass + render + quantise on the device + copy of the RGB8 image to pinned host memory + "
                                       "synchronise (ms_same_camera: the tile pass cached; ms_with_fp32_radiance_copy: same camera, the "
                                       "9.4 MB fp32 radiance copied instead, the figure of rounds 1-2)"}

    # ---- CPU baseline (rank 0, N = 1 only): bounded sample of the same workload ---------------------
    cpu_note = ""
    if world == 1 and args.cpu_col_stride != 0:
        from oracle import pyoracle  # the checker, used here only as the timed CPU baseline
        n_threads = args.cpu_threads or len(os.sched_getaffinity(0))  # every core this process may use
        if args.cpu_col_stride < 0:
            # Default: a 64th of the frame's columns is timed first; then the largest power-of-two fraction of the columns
            # whose estimate fits --cpu-budget-s (120 s) is timed, and the line says which. (Round 3 timed the whole frame
            # wherever 64 cores were to be had: 287 s of a 293-s run on 256 threads, 13 s short of its own limit.)
            args.cpu_col_stride = 1
            t0 = time.perf_counter()
            pyoracle.render(cam, host_scene, abi.default_opts(spp=spp, seed=args.seed), n_threads=n_threads, want_rgb8=False, col_stride=64)
            est_full_s = (time.perf_counter() - t0) * 64.0
            while est_full_s / args.cpu_col_stride > args.cpu_budget_s and args.cpu_col_stride < 64:
                args.cpu_col_stride *= 2
            if args.cpu_col_stride > 1:
                cpu_note = (f"; the whole frame was estimated at {est_full_s:.0f} s on this host: the largest power-of-two fraction of "
                            f"the columns that fits {args.cpu_budget_s:.0f} s is timed")
        cols = len(range(0, W, args.cpu_col_stride))
        t0 = time.perf_counter()
        rad, _, cpu_rays = pyoracle.render(cam, host_scene, abi.default_opts(spp=spp, seed=args.seed),
                                           n_threads=n_threads, want_rgb8=False, col_stride=args.cpu_col_stride)
        cpu_s = time.perf_counter() - t0
        cpu_samples = cols * H * spp
        out["cpu_baseline"] = {
            "value": round(cpu_samples / cpu_s / 1e6, 4), "unit": "Mray-samples/s", "cores": n_threads,
            "host_logical_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{'every column' if args.cpu_col_stride == 1 else 'every 2nd column' if args.cpu_col_stride == 2 else f'every {args.cpu_col_stride}th column'} ({cols} of {W}) x {H} rows x {spp} spp of the same "
                      f"scene and seed, {cpu_s:.1f} s; C++ restatement of the reference AVX path (brute force over "
                      f"all triangles), not the Rust binary{cpu_note}",
        }
        if args.check:
            import numpy as np
            gpu = image.cpu().numpy()[:, ::args.cpu_col_stride]
            cpu = rad[:, ::args.cpu_col_stride]
            out["cpu_baseline"]["gpu_matches_bitwise"] = bool(np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32)))
            out["cpu_baseline"]["rmse"] = float(np.sqrt(np.mean((gpu.astype(np.float64) - cpu) ** 2)))
    if dbg is not None:
        out["debug_counters"] = dbg
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
