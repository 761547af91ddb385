#!/bin/bash
set -o pipefail
O=gpurun_out/r04s; mkdir -p $O
for lpr in 8 16 4; do
  FCVM_ASM_LPR=$lpr timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "straight_into" > $O/tests_lpr$lpr.log 2>&1; rc=$?; echo "LPR=$lpr tests rc=$rc"; tail -2 $O/tests_lpr$lpr.log
  [ $rc -eq 0 ] || { tail -30 $O/tests_lpr$lpr.log; exit 1; }
done
for lpr in 4 8 16; do
  FCVM_ASM_LPR=$lpr ROUTES=direct python tools/sessions/asm_time_r04.py 2>&1 | grep calc_gsm | sed "s/^/LPR=$lpr /"
  FCVM_ASM_LPR=$lpr ROUTES=direct python tools/sessions/asm_time_r04.py tangent 2>&1 | grep calc_gsm | sed "s/^/LPR=$lpr /"
done
