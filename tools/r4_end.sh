#!/bin/bash
# Round 4, the end of a launch: parity of taper + spread, then the knob sweep (tools/endsweep.py). usage: tools/r4_end.sh TAG variants...
set -o pipefail
TAG=$1; shift
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "${KEXPR:-merging or shared or pipelined or cfg1 or streamed}" > $O/pytest_$TAG.log 2>&1
rc=$?; tail -3 $O/pytest_$TAG.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
if [ -n "$BASELIB" ]; then
  RBRT_HIP_LIB=$BASELIB timeout -k 10 300 python3 tools/endsweep.py --rounds ${ROUNDS:-3} ${SWEEP_ARGS:-} "-" > $O/sweep_${TAG}_base.log 2>&1 || exit 1
  tail -1 $O/sweep_${TAG}_base.log
fi
timeout -k 10 ${SWEEP_TMO:-900} python3 tools/endsweep.py --rounds ${ROUNDS:-3} ${SWEEP_ARGS:-} "$@" > $O/sweep_$TAG.log 2>&1
rc=$?; tail -${TAILN:-20} $O/sweep_$TAG.log; echo "sweep rc=$rc"
