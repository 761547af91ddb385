#!/bin/bash
set -o pipefail
O=gpurun_out/r04l; mkdir -p $O
FCVM_PATTERN_TIMING=1 FCVM_SETUP_TIMING=1 python tools/setup_profile.py --plate-h 1.45 > $O/setup_stages.txt 2>&1
grep -v "^ \|ncalls\|function calls\|Ordered\|List reduced" $O/setup_stages.txt | head -70
