#!/bin/bash
# round 5, verdict item 3: does the placement of the SpMV's x / y vectors set its time?
set -o pipefail
O=gpurun_out/r05k; mkdir -p $O
timeout -k 10 400 python3 tools/sessions/spmv_vecplace_r05.py 2>&1 | grep "^instance" | tee $O/vecplace.txt
