#!/bin/bash
O=gpurun_out/r03x
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python3 tools/gnly_probe.py --steps 4 > $O/gnly.log 2>&1; tail -2 $O/gnly.log
FCVM_REPACK_OVERLAP=0 timeout -k 10 300 python3 tools/gnly_probe.py --steps 4 > $O/gnly_nooverlap.log 2>&1; tail -1 $O/gnly_nooverlap.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "reruns or gnly or assemble or window_format or 16bit or two_right" > $O/tests.log 2>&1; echo rc=$?; tail -4 $O/tests.log
