#!/bin/bash
# round-2 final GPU session: the whole -m gpu suite, smoke, default bench line, kernel statistics of the bench command
O=gpurun_out/r02fin
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/suite.log 2>&1; echo "suite rc $?"
tail -14 $O/suite.log | cut -c1-160
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log | cut -c1-300
timeout -k 10 600 python3 bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r02fin/bench_default.log") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup", "pcg_iterations_per_step")})
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_ms", "back_to_back_launch_ms", "stream_read_GBs_this_box")})
print("assembly", d["assembly"], "setup", d["setup_s"])
print("plate analysis_s", d["north_star_workload"]["analysis_s"], "pattern", d["north_star_workload"]["pattern_and_upload_s"])
print("same mesh gpu/cpu", d["cpu_baseline"]["same_mesh"]["gpu_over_cpu_newton_iterations_per_s"])
PY
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-plate > $GRAFT_REPO_ROOT/$O/bench_trace.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -16 "$f" | cut -c1-140
