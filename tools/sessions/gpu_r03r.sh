#!/bin/bash
O=gpurun_out/r03r
mkdir -p $O
export TMPDIR=/tmp
FCVM_PATTERN_TIMING=1 FCVM_VERBOSE=1 timeout -k 10 200 python3 tools/setup_profile.py --plate-h 1.45 > $O/setup_profile.log 2>&1; echo rc=$?
grep -v "^ \|^$" $O/setup_profile.log | head -50
