for ws in 0 4 16 64 256 1024; do
export RBRT_HIP_LAB=1  # the scheduling knobs below are lab knobs (include/rbrt_hip_debug.h)
  r=$(RBRT_WORK_STRIPES=$ws timeout -k 10 120 python3 bench.py --steps 16 --warmup 3 --isolated-steps 8 --cpu-col-stride 0 --single-frames 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline'].get('isolated_leg',{}).get('kernel_ms'), j['config']['image_sha256_16'])")
  echo "stripes=$ws full -> $r"
  r=$(RBRT_WORK_STRIPES_OVERLAP=$ws timeout -k 10 120 python3 bench.py --steps 80 --warmup 8 --emulate-rank-of 8 --cpu-col-stride 0 --single-frames 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'])")
  echo "stripes_short=$ws r8 -> $r"
done
