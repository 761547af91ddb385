#!/bin/bash
# round-2 GPU session W: fused dots in the Newton driver; default bench line (with CPU baseline and the north-star plate run)
O=gpurun_out/r02w
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "not elastica and not bench_line" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench_short.log 2>&1 || { tail $O/bench_short.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*' $O/bench_short.log
( time timeout -k 10 900 python3 bench.py > $O/bench_default.log 2>$O/bench_default.err ) 2> $O/bench_default.time || { tail $O/bench_default.err; exit 1; }
cat $O/bench_default.time | tail -3
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r02w/bench_default.log") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "pcg_iterations_per_step")})
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_ms", "back_to_back_launch_ms", "stream_read_GBs_this_box")})
print("assembly", d["assembly"], "setup", d["setup_s"])
print("cpu", {k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
print("plate", d.get("north_star_workload"))
PY
