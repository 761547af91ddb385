#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "assemble or straight_into or switches or size_independent" > gpurun_out/r04zk_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04zk_tests.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
for mode in elastic tangent; do
  arg=""; [ $mode = tangent ] && arg=tangent
  ROUTES=direct timeout -k 10 300 python3 tools/sessions/asm_time_r04.py $arg > gpurun_out/r04zk_$mode.log 2>&1
  grep calc_gsm gpurun_out/r04zk_$mode.log | cut -c1-100
done
timeout -k 10 300 python tools/gnly_probe.py --steps 6 2>&1 | grep "^iteration" | tail -3 | cut -c1-150
