for cfg in "0 0" "8 3" "8 4" "6 4" "5 4" "4 4" "4 5" "4 6" "3 6" "3 8" "2 8" "6 3" "5 5"; do
  set -- $cfg
  if [ "$1" = "0" ]; then w=""; else w="RBRT_WAVES_PER_CU=$1"; fi
  r=$(env $w timeout -k 10 120 python3 bench.py --steps 60 --warmup 8 --cpu-col-stride 0 --emulate-rank-of 8 --pipeline $2 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['roofline']['kernel_ms_pipelined'])")
  echo "waves/cu=$1 pipeline=$2 -> $r"
done
