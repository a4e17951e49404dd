#!/bin/bash
# Round 4 session c: what the end of a launch is made of. (1) counters of the spread; (2) the eighth against pipeline depth x grid
# size; (3) the tail against the depth limit (diagnosis); (4) region timers of the eighth with and without spreading.
set -o pipefail
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python3 tools/endsweep.py --rounds 1 --worlds 1,8 "STATS=1" "STATS=1 RBRT_SPREAD_MIN=16" "STATS=1 RBRT_SPREAD_MIN=16 RBRT_SPREAD_TAIL=0" "STATS=1 RBRT_TAPER_CHUNKS=2" > $O/c_stats.log 2>&1 || { tail -5 $O/c_stats.log; exit 1; }
grep "^stats" $O/c_stats.log
timeout -k 10 900 python3 tools/endsweep.py --rounds 3 --worlds 8 --steps 40 "-" "PIPE=4" "PIPE=4 RBRT_WAVES_PER_CU=8" "PIPE=4 RBRT_WAVES_PER_CU=4" "PIPE=6 RBRT_WAVES_PER_CU=4" "PIPE=8 RBRT_WAVES_PER_CU=4" "PIPE=8 RBRT_WAVES_PER_CU=2" "PIPE=6 RBRT_WAVES_PER_CU=8" "PIPE=3 RBRT_WAVES_PER_CU=6" "PIPE=5 RBRT_WAVES_PER_CU=5" "PIPE=2 RBRT_WAVES_PER_CU=8" "PIPE=3 RBRT_WAVES_PER_CU=16" > $O/c_pipe.log 2>&1 || { tail -5 $O/c_pipe.log; exit 1; }
sed -n '/summary/,$p' $O/c_pipe.log
timeout -k 10 600 python3 tools/endsweep.py --rounds 2 --worlds 1,8 "-" "DEPTH=24" "DEPTH=12" "DEPTH=6" "DEPTH=3" > $O/c_depth.log 2>&1 || { tail -5 $O/c_depth.log; exit 1; }
sed -n '/summary/,$p' $O/c_depth.log
RBRT_HIP_LIB=rbrt_amd/lib/librbrt_hip_timers.so timeout -k 10 300 python3 tools/region_profile.py --emulate-rank-of 8 --frames 6 > $O/c_regions_r8.txt 2>&1 || { tail -5 $O/c_regions_r8.txt; exit 1; }
RBRT_HIP_LAB=1 RBRT_SPREAD_MIN=16 RBRT_HIP_LIB=rbrt_amd/lib/librbrt_hip_timers.so timeout -k 10 300 python3 tools/region_profile.py --emulate-rank-of 8 --frames 6 > $O/c_regions_r8_spread.txt 2>&1 || { tail -5 $O/c_regions_r8_spread.txt; exit 1; }
grep "wall clock\|summed drain\|longest drain" $O/c_regions_r8.txt $O/c_regions_r8_spread.txt
echo "session c done"
