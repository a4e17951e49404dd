# drain hand-over sweep: group size x threshold x collector priority; full frame and an eighth
run() { # env...
  for emu in 0 8; do
    if [ $emu = 0 ]; then a="--steps 16 --warmup 3 --isolated-steps 6"; else a="--steps 60 --warmup 8 --emulate-rank-of 8"; fi
    r=$(env "$@" timeout -k 10 120 python3 bench.py $a --cpu-col-stride 0 --single-frames 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline'].get('isolated_leg',{}).get('kernel_ms'), j['config']['image_sha256_16'])")
    echo "$* emu=$emu -> $r"
  done
}
run RBRT_XGROUP=0
for g in 2 4 8 16; do for t in 32 64 256; do run RBRT_XGROUP=$g RBRT_XTHRESH=$t; done; done
run RBRT_XGROUP=8 RBRT_XTHRESH=32 RBRT_DRAIN_MODE=5
run RBRT_XGROUP=16 RBRT_XTHRESH=32 RBRT_DRAIN_MODE=5
run RBRT_XGROUP=16 RBRT_XTHRESH=16 RBRT_DRAIN_MODE=5
run RBRT_XGROUP=4 RBRT_XTHRESH=256 RBRT_DRAIN_MODE=5
