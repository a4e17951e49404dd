#!/bin/bash
# What the rbrt CLI reports for a handful of workloads (GPU box): render seconds and Mray-samples/s from --report.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/r4/cli; mkdir -p $O
(cd $ROOT && python3 -m rbrt_amd.standin $O/bunny.obj > /dev/null 2>&1)
sed "s#obj_filepath: bunny.obj#obj_filepath: $O/bunny.obj#" $ROOT/scenes/example_scene.yaml > $O/scene.yaml
R=$ROOT/rbrt_amd/bin/rbrt
run() { local tag=$1; shift; $R --config $O/scene.yaml -t $O/out_$tag.png --report $O/rep_$tag.json "$@" > $O/log_$tag.txt 2>&1 || { echo "$tag FAILED"; tail -3 $O/log_$tag.txt; return; }
  python3 -c "
import json; j=json.load(open('$O/rep_$tag.json')); print('$tag', 'render_s', j['render_s'], 'Mray/s', round(j['mray_samples_per_s'],1), 'passes', j.get('passes'), 'total_s', j['total_s'])"; }
run a50 --height 768 --width 1024 --samples 50
run a50b --height 768 --width 1024 --samples 50
run a500 --height 768 --width 1024 --samples 500
run a500p --height 768 --width 1024 --samples 500 --pass-samples 50
run hd512 --height 1080 --width 1920 --samples 512
run g4 --height 768 --width 1024 --samples 200 --gpus 4 --oversubscribe
run big --height 2048 --width 2048 --samples 64
