#!/bin/bash
# Round 5, session 7: what the helper launches do at the end of a 20-step stream: the watcher's log and a kernel trace.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5/s7; rm -rf $OUT; mkdir -p $OUT
B="bench.py --warmup 5 --cpu-col-stride 0 --single-frames 0 --one-shot 0 --isolated-steps 0 --same-camera-steps 0 --steps 20"
RBRT_HIP_LAB=1 RBRT_TRACE_LAUNCHES=1 timeout -k 10 300 python3 $B > $OUT/trace_on.json 2> $OUT/trace_on.err
grep -c "helper launch" $OUT/trace_on.err; grep "helper launch" $OUT/trace_on.err | tail -12
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_on -- python3 $B > $OUT/kt_on.json 2> $OUT/kt_on.err
RBRT_HIP_LAB=1 RBRT_HELPERS=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_off -- python3 $B > $OUT/kt_off.json 2> $OUT/kt_off.err
for t in on off; do f=$(ls $OUT/kt_$t/*/*kernel_trace.csv | head -1); python3 tools/trace_timeline.py $f > $OUT/timeline_$t.txt; wc -l $OUT/timeline_$t.txt; done
echo "session done"
