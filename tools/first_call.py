"""First render call of a scene against the later ones (GPU box): what a one-shot caller such as the rbrt CLI pays."""
import os, sys, time, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent; sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import rbrt_amd
from rbrt_amd import abi, standin
import ctypes as C
work = Path(tempfile.mkdtemp())
obj = standin.ensure_obj(work / "bunny.obj", 69451, "smooth")
(work / "scene.yaml").write_text((ROOT / "scenes" / "example_scene.yaml").read_text().replace("obj_filepath: bunny.obj", f"obj_filepath: {obj}"))
hs = abi.HostScene(work / "scene.yaml", 768, 1024)
lib = rbrt_amd.load_hip()
import torch
def sync(): torch.cuda.synchronize()
for pipe in (3, 1, 0):
    t0 = time.perf_counter(); scene = rbrt_amd.HipScene(hs); sync(); t1 = time.perf_counter()
    if pipe: scene.set_pipeline(pipe)
    sync(); t2 = time.perf_counter()
    buf = torch.empty(1024*768*3, dtype=torch.uint8, device="cuda")
    ts = []
    for k in range(4):
        a = time.perf_counter()
        scene.render_device(hs.camera, abi.default_opts(spp=50, seed=1), None, buf.data_ptr(), None); sync()
        ts.append((time.perf_counter() - a) * 1e3)
    print(f"pipeline {pipe}: create {1e3*(t1-t0):.1f} ms, set_pipeline {1e3*(t2-t1):.1f} ms, renders {[round(x,2) for x in ts]}")
    scene.close()
