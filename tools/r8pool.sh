for cfg in "RBRT_POOL=128" "RBRT_POOL=96" "RBRT_POOL=160" "RBRT_POOL=96 RBRT_Y_LOW=20" "RBRT_POOL=128 RBRT_Y_LOW=20" "RBRT_POOL=128 RBRT_LEAF_ROUND=4" "RBRT_POOL=128 RBRT_SHADE_ROUNDS=2" "RBRT_POOL=128 RBRT_SHADE_ROUNDS=4"; do
  r=$(env $cfg timeout -k 10 120 python3 bench.py --steps 80 --warmup 8 --cpu-col-stride 0 --emulate-rank-of 8 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline']['kernel_ms_pipelined'])")
  echo "$cfg -> $r"
done
