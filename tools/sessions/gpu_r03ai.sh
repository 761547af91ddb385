#!/bin/bash
# round 3: stage times of the hierarchy set-up; full GPU suite after the device element map
O=gpurun_out/r03ai
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
FCVM_SETUP_TIMING=1 timeout -k 10 200 python3 tools/setup_inside.py > $O/setup.txt 2>&1; grep "fcvm set-up\|RiksDriver" $O/setup.txt
timeout -k 10 800 python3 -m pytest tests -q -m gpu --durations=6 > $O/gpu_tests.log 2>&1; echo tests rc=$?; tail -12 $O/gpu_tests.log
