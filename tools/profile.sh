#!/bin/bash
# Profiles the default bench.py workload on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the ISOLATED run (--pipeline 1: launches do not overlap, so the CSV's
#      average duration of trace_megakernel is the roofline's denominator) and of the default (pipelined) command
#   2. separate --pmc passes (counters only, as gpurun requires): SQ activity, then the memory-side
#      traffic counters (FETCH_SIZE and WRITE_SIZE need a pass each: MI355X_MICROARCH.md "rocprofv3 PMC slots")
# Output under gpurun_out/profile/; tools/summarize_profile.py <tag> turns it into profiles/<tag>_*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile
rm -rf "$OUT" && mkdir -p "$OUT"
# what is being profiled: the hash bench.py later checks the counters' freshness against (rbrt_amd/srchash.py)
python3 -c "from rbrt_amd.srchash import kernel_source_sha256 as h; print(h())" > "$OUT/kernel_source_sha256.txt" || exit 1
COMMON="--cpu-col-stride 0 --single-frames 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_isolated" -- python3 bench.py --steps 10 --warmup 2 --pipeline 1 $COMMON \
    > "$OUT/bench_stats_isolated.json" 2> "$OUT/bench_stats_isolated.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -- python3 bench.py --steps 10 --warmup 2 --isolated-steps 0 $COMMON \
    > "$OUT/bench_stats_default.json" 2> "$OUT/bench_stats_default.err" || exit 1
run_pmc() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 0 --pipeline 1 $COMMON \
      > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || { echo "pmc pass $name failed"; tail -3 "$OUT/bench_$name.err"; }
}
run_pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run_pmc sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY
# the DYNAMIC instruction mix (round 4: the issue model had priced the executed VALU instructions with the static mix of the
# binary): executed VALU instructions by class, scalar-memory and branch instructions, LDS by kind
run_pmc mix1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
run_pmc mix2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_WR SQ_INSTS_VSKIPPED
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
run_pmc l1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
echo "profile done"
