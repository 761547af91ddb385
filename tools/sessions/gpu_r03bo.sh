#!/bin/bash
# round 3: fixed-point conversion with multiplications -- parity subset, then the bench (sustained clocks)
O=gpurun_out/r03bo
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 400 python3 -m pytest tests -q -m gpu -x -k "symmetric_window or window_format_spmv or lockstep or reruns or 16bit" > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -4 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 bench.py --no-cpu --no-cube > $O/bench.json 2> $O/bench.err; echo bench rc=$?
python3 -c "
import json
d=json.loads(open('$O/bench.json').read()); print(round(d['ms_per_step'],2), d['pcg_iterations_per_solve'], round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['back_to_back_launch_ms'],4), round(d['north_star_workload']['analysis_s'],3))"
