#!/bin/bash
O=gpurun_out/r03v
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
FCVM_PATTERN_TIMING=1 timeout -k 10 200 python3 tools/setup_profile.py --plate-h 1.45 > $O/setup_profile.log 2>&1; echo rc=$?
grep "set_mesh\|window format:\|GpuModel\|first calc\|build_level2" $O/setup_profile.log | head -20
timeout -k 10 300 python3 tools/gnly_probe.py --steps 4 > $O/gnly.log 2>&1; tail -2 $O/gnly.log
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read()); print(d['ms_per_step'], d['pcg_iterations_per_solve'], d['roofline']['frac'], d['setup_s'], d['north_star_workload']['analysis_s'], d['config']['preconditioner'][:90])"
