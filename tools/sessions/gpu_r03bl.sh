#!/bin/bash
# round 3: symmetric form built on its own thread -- parity subset, set-up time
O=gpurun_out/r03bl
mkdir -p $O
timeout -k 10 400 python3 -m pytest tests -q -m gpu -x -k "symmetric_window or window_format_spmv or reruns or set_up_switches or element_map" > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -4 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
FCVM_PATTERN_TIMING=1 timeout -k 10 200 python3 tools/sessions/gm_prof_r03.py > $O/gm.txt 2>&1; grep "GpuModel\|set_mesh:\|symmetric form" $O/gm.txt | tail -12
