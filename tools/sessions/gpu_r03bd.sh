#!/bin/bash
O=gpurun_out/r03bd
mkdir -p $O
FCVM_PATTERN_TIMING=1 timeout -k 10 300 python3 tools/sessions/gm_prof_r03.py > $O/prof.txt 2>&1; echo rc=$?; grep -v "^W2026\|^$\|fcvm pattern:\|fcvm window" $O/prof.txt | tail -45 | cut -c1-140
