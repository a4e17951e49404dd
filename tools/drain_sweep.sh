for dm in 1 9 1 9; do
  for emu in 0 8; do
    if [ $emu = 0 ]; then a="--steps 16 --warmup 3 --isolated-steps 6"; else a="--steps 80 --warmup 8 --emulate-rank-of 8"; fi
    r=$(RBRT_DRAIN_MODE=$dm timeout -k 10 120 python3 bench.py $a --cpu-col-stride 0 --single-frames 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline'].get('isolated_leg',{}).get('kernel_ms'), j['config']['image_sha256_16'])")
    echo "drain_mode=$dm emu=$emu -> $r"
  done
done
