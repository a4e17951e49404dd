#!/bin/bash
# The driver's 20-step command ten times with elastic launches on and off in turn, one process each (run on the GPU box):
# the distribution of ms per step, which a single run does not show (profiles/r05_helpers_newest_first.txt).
mkdir -p gpurun_out/r5/rep
E="python3 bench.py --gpus 1 --warmup 5 --cpu-col-stride 0 --single-frames 0 --one-shot 0 --isolated-steps 0 --same-camera-steps 0 --steps 20"
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 100 $E > gpurun_out/r5/rep/on_$i.json 2>/dev/null || exit 1
  RBRT_HIP_LAB=1 RBRT_HELPERS=0 timeout -k 10 100 $E > gpurun_out/r5/rep/off_$i.json 2>/dev/null || exit 1
done
python3 - <<'PY'
import json,glob
for k in ("on","off"):
    v=[json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob(f"gpurun_out/r5/rep/{k}_*.json"))]
    print(k, " ".join(f"{x:.3f}" for x in v), " median", sorted(v)[len(v)//2])
PY
