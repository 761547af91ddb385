#!/bin/bash
O=gpurun_out/r02zb
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "sparse_aggregate or pcg or precond" > $O/parity.log 2>&1; tail -3 $O/parity.log | cut -c1-220
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
FCVM_VERBOSE=1 timeout -k 10 420 python3 bench.py --cells 110 --steps 4 --warmup 2 --no-cpu --no-plate > $O/bench8m.log 2> $O/bench8m.err || { tail -5 $O/bench8m.err; exit 1; }
grep "sparse aggregate" $O/bench8m.err | head -2
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"assembly_plus_preconditioner_hierarchy": [0-9.]*\|aggregates x[^"]*' $O/bench8m.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_configs.py -x -q -k "config4" > $O/config4.log 2>&1; tail -3 $O/config4.log | cut -c1-200
