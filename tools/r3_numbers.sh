#!/bin/bash
# Round-3 reference numbers on one box: default bench, rough stand-in, 871k mesh, an eighth of the frame.
O=gpurun_out/r3; mkdir -p $O
C="--cpu-col-stride 0"
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "product_host or cli" --timeout 300 > $O/pytest_host.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_host.log
timeout -k 10 300 python3 bench.py $C > $O/bench_smooth.json 2> $O/bench_smooth.err; echo "smooth rc=$?"
RBRT_BENCH_DEBUG=1 timeout -k 10 300 python3 bench.py $C --mesh rough > $O/bench_rough.json 2> $O/bench_rough.err; echo "rough rc=$?"
RBRT_BENCH_DEBUG=1 timeout -k 10 300 python3 bench.py $C --steps 4 --warmup 1 > $O/bench_smooth_dbg.json 2> $O/bench_smooth_dbg.err
timeout -k 10 300 python3 bench.py $C --triangles 871414 --single-frames 0 > $O/bench_dragon.json 2> $O/bench_dragon.err; echo "dragon rc=$?"
for n in 2 4 8; do timeout -k 10 300 python3 bench.py $C --steps 40 --warmup 5 --emulate-rank-of $n > $O/bench_r$n.json 2> $O/bench_r$n.err; done
python3 - <<'PY'
import json
for n in ("smooth","rough","dragon","r2","r4","r8"):
    try:
        j=json.loads(open(f"gpurun_out/r3/bench_{n}.json").read().strip().splitlines()[-1])
        r=j["roofline"]; c=r["counters"]
        print(n, "value", j["value"], "ms/step", j["ms_per_step"], "iso", r["kernel_ms"], "frac", r["frac"], "alg GB", round(r["algorithmic_bytes_per_launch"]/1e9,2),
              "nodes", c["nodes_visited"], "tris", c["tris_tested"], "rays", c["rays"], "single", j.get("single_frame",{}).get("ms"), "sha", j["config"]["image_sha256_16"], "issue", r.get("issue_mix",{}).get("frac"))
    except Exception as e:
        print(n, "failed", e)
PY
