# usage: bash tools/pipe_test.sh WORLD -- per-step time of rank 0's share for (waves per CU, pipeline depth) pairs
export RBRT_HIP_LAB=1  # the scheduling knobs below are lab knobs (include/rbrt_hip_debug.h)
w=${1:-8}
for cfg in "16 2" "16 3" "12 3" "10 3" "8 2" "8 3" "8 4" "6 3" "6 4" "5 3" "5 4" "4 4"; do
set -- $cfg
RBRT_WAVES_PER_CU=$1 python3 bench.py --steps 24 --warmup 6 --cpu-col-stride 0 --emulate-rank-of $w --pipeline $2 > gpurun_out/p.json 2>gpurun_out/p.err || { tail -5 gpurun_out/p.err; exit 1; }
python3 -c "import json,sys; j=json.load(open('gpurun_out/p.json')); print('world', sys.argv[1], 'waves/CU', sys.argv[2], 'pipeline', sys.argv[3], 'ms_per_step', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'])" $w $1 $2
done
