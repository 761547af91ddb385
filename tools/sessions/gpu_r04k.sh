#!/bin/bash
set -o pipefail
O=gpurun_out/r04k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "straight_into or tangent or calcGSM or gnly or buckling" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
ROUTES=direct python tools/sessions/asm_time_r04.py tangent 2>&1 | grep calc_gsm
ROUTES=direct python tools/sessions/asm_time_r04.py 2>&1 | grep calc_gsm
python tools/gnly_probe.py --cells 55 --steps 4 2>&1 | grep iteration
