#!/bin/bash
# round 3: triangular inverse by recursive doubling -- parity with LAPACK, set-up stage time, level-2 tests
O=gpurun_out/r03ak
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python3 -m pytest tests -q -m gpu -x -k "dense_spd or sparse_aggregate or preconditioners or pcg_solve" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
FCVM_SETUP_TIMING=1 timeout -k 10 200 python3 tools/setup_inside.py > $O/setup.txt 2>&1; grep "fcvm set-up: level 2\|RiksDriver" $O/setup.txt
FCVM_TRINV_ROWS=1 FCVM_SETUP_TIMING=1 timeout -k 10 200 python3 tools/setup_inside.py > $O/setup_rows.txt 2>&1; grep "dense SPD\|RiksDriver" $O/setup_rows.txt
