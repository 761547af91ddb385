#!/bin/bash
O=gpurun_out/r03i
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1100 python3 -m pytest tests -q -m gpu --durations=12 > $O/gpu_tests.log 2>&1
echo tests rc=$?; tail -30 $O/gpu_tests.log
