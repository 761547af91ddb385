#!/bin/bash
# round 3: full GPU suite after the lazy allocation of the full-form FP64 window values
O=gpurun_out/r03bn
mkdir -p $O
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 800 python3 -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo tests rc=$?; tail -4 $O/gpu_tests.log
