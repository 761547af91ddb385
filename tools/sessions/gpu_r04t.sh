#!/bin/bash
set -o pipefail
O=gpurun_out/r04t; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "straight_into" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || { tail -30 $O/tests.log; exit 1; }
for cfg in "4 28" "8 20" "8 28" "8 40" "8 53" "16 40" "16 53"; do
  set -- $cfg
  FCVM_ASM_LPR_LONG=$1 FCVM_ASM_LPR_FROM=$2 ROUTES=direct python tools/sessions/asm_time_r04.py 2>&1 | grep calc_gsm | sed "s/^/long=$1 from>$2 /" | cut -c1-110
  FCVM_ASM_LPR_LONG=$1 FCVM_ASM_LPR_FROM=$2 ROUTES=direct python tools/sessions/asm_time_r04.py tangent 2>&1 | grep calc_gsm | sed "s/^/long=$1 from>$2 /" | cut -c1-110
done
