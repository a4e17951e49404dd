# usage: bash tools/pipe_test.sh  -- per-step time for (waves per CU, pipeline depth) pairs, rank-0 share of world 8 and world 1
for w in 8 1; do for cfg in "16 1" "16 2" "8 2" "8 3" "6 3" "5 3" "4 4" "10 2" "12 2"; do
set -- $cfg
RBRT_WAVES_PER_CU=$1 python3 bench.py --steps 16 --warmup 4 --cpu-col-stride 0 --emulate-rank-of $w --pipeline $2 > gpurun_out/p.json 2>gpurun_out/p.err || { tail -5 gpurun_out/p.err; exit 1; }
python3 -c "import json,sys; j=json.load(open('gpurun_out/p.json')); print('world', sys.argv[1], 'waves/CU', sys.argv[2], 'pipeline', sys.argv[3], 'ms_per_step', j['ms_per_step'], 'value', j['value'], 'kernel_ms', j['roofline']['kernel_ms'])" $w $1 $2
done; done
