#!/bin/bash
set -o pipefail
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "straight_into" > $O/tests_direct.log 2>&1; rc=$?; tail -3 $O/tests_direct.log
[ $rc -eq 0 ] || exit 1
ROUTES_LIST=direct bash tools/sessions/gpu_r04f.sh 2>&1 | grep -v "simple_timer\|rocprofv3\]"
ROUTES_LIST=direct ASM_ARGS=tangent bash tools/sessions/gpu_r04f.sh 2>&1 | grep -v "simple_timer\|rocprofv3\]" | grep calc_gsm
ROUTES_LIST=copy ASM_ARGS=tangent bash tools/sessions/gpu_r04f.sh 2>&1 | grep -v "simple_timer\|rocprofv3\]" | grep calc_gsm
