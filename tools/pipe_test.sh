for w in 8 1; do for pl in 1 2 3; do
python3 bench.py --steps 16 --warmup 4 --cpu-col-stride 0 --emulate-rank-of $w --pipeline $pl > gpurun_out/p.json 2>gpurun_out/p.err || { tail -5 gpurun_out/p.err; exit 1; }
python3 -c "import json,sys; j=json.load(open('gpurun_out/p.json')); print('world', sys.argv[1], 'pipeline', sys.argv[2], 'ms_per_step', j['ms_per_step'], 'value', j['value'], 'kernel_ms', j['roofline']['kernel_ms'])" $w $pl
done; done
