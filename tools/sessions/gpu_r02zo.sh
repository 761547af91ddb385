#!/bin/bash
O=gpurun_out/r02zo
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/gnly_probe.py > $O/gnly.log 2>&1; tail -4 $O/gnly.log | cut -c1-200
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q --durations=5 -k "gnly or GNLY or buckling or u100 or reference_documents or geometric" > $O/gnly_tests.log 2>&1; tail -9 $O/gnly_tests.log | cut -c1-160
