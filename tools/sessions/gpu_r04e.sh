#!/bin/bash
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "straight_into or two_right_hand" > $O/tests_direct.log 2>&1; rc=$?; tail -3 $O/tests_direct.log
[ $rc -eq 0 ] || exit 1
bash tools/sessions/gpu_r04d.sh 2>&1 | grep "calc_gsm"
