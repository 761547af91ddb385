#!/bin/bash
O=gpurun_out/r03n
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "two_right_hand_sides" > $O/tests.log 2>&1
echo tests rc=$?; tail -25 $O/tests.log
FCVM_SOLVE2=0 timeout -k 10 300 python3 tools/gnly_probe.py > $O/gnly_seq.log 2>&1; echo seq rc=$?; tail -4 $O/gnly_seq.log
timeout -k 10 300 python3 tools/gnly_probe.py > $O/gnly_pair.log 2>&1; echo pair rc=$?; tail -4 $O/gnly_pair.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q --durations=8 -k "gnly or GNLY or buckling or u100 or elastica or barbuckling" > $O/gnly_tests.log 2>&1; echo rc=$?; tail -14 $O/gnly_tests.log
