#!/bin/bash
# round 3: two ranks of the REAL per-rank size (2 x 1.0 M tet10) sharing one GPU over gloo -- does the N > 1 path hold up at that
# size (set-up time, memory)?  The absolute step time is not meaningful (host-staged exchanges, one GPU for both ranks).
O=gpurun_out/r03ba
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
FCVM_DIST_SHARE_GPU=1 timeout -k 10 1000 python3 bench.py --gpus 2 --steps 3 --warmup 2 > $O/bench2.json 2> $O/bench2.err; echo bench2 rc=$?; tail -c 1800 $O/bench2.json; tail -5 $O/bench2.err | cut -c1-300
