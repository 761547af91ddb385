#!/bin/bash
O=gpurun_out/r03az
mkdir -p $O
timeout -k 10 300 python3 tools/newton_host_profile.py > $O/prof.txt 2>&1; echo rc=$?; grep -v "^W2026\|^$" $O/prof.txt | head -75 | cut -c1-150
