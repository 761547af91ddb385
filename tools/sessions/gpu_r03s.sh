#!/bin/bash
# round 3, call s: full GPU suite after the solver / dist changes, 2-rank rehearsal of bench.py --gpus 2 on one GPU (gloo)
O=gpurun_out/r03s
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
FCVM_DIST_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --cells 16 --steps 3 --warmup 1 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo dist bench rc=$?; tail -c 400 $O/bench_2ranks.err; head -c 900 $O/bench_2ranks.json; echo
timeout -k 10 1000 python3 -m pytest tests -q -m gpu --durations=10 > $O/gpu_tests.log 2>&1
echo tests rc=$?; tail -22 $O/gpu_tests.log
