#!/bin/bash
# round-2 GPU session N: the whole -m gpu suite with per-test durations (a heartbeat file keeps the silence watchdog informed
# while a long test runs; every step is bounded by timeout), then the clock / power log of the SpMV
O=gpurun_out/r02n
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -v --durations=0 > $O/suite.log 2>&1; echo "suite rc $?"
grep -E "passed|failed|error" $O/suite.log | tail -3
grep -E "^[0-9.]+s (call|setup)" $O/suite.log | head -25
