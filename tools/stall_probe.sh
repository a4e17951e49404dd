#!/bin/bash
# Does this box stall the one-shot call (20-25 ms in whichever HIP call comes next), and does polling for completion signals
# (HSA_ENABLE_INTERRUPT=0) change that? bench.py's one_shot leg four times each, in turn (run on the GPU box through gpurun).
mkdir -p gpurun_out/r5/stall
E="python3 bench.py --steps 5 --warmup 2 --cpu-col-stride 0 --single-frames 0 --same-camera-steps 0 --isolated-steps 0 --one-shot 7"
for i in 1 2 3 4; do
  for v in irq poll; do
    if [ $v = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
    timeout -k 10 100 $E > gpurun_out/r5/stall/${v}_$i.json 2>/dev/null || exit 1
    python3 -c "
import json; o=json.loads(open('gpurun_out/r5/stall/${v}_$i.json').read().strip().splitlines()[-1])['one_shot']; print('$v $i', o['ms'], o['ms_min'], o['ms_max'])"
  done
done
