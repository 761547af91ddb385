#!/bin/bash
# round 4: where the one-time set-up inside an analysis goes (FCVM_SETUP_TIMING / FCVM_PATTERN_TIMING), 1 M plate
mkdir -p gpurun_out
FCVM_SETUP_TIMING=1 FCVM_PATTERN_TIMING=1 timeout -k 10 300 python3 tools/setup_inside.py > gpurun_out/r04zf_setup.log 2>&1
grep -v "^ \|^$" gpurun_out/r04zf_setup.log | cut -c1-200 | head -80
