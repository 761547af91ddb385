#!/bin/bash
# round 5, verdict item 3: cold address translation in front of the in-PCG SpMV?
set -o pipefail
O=gpurun_out/r05j; mkdir -p $O
timeout -k 10 400 python3 tools/sessions/spmv_touch_r05.py 2>&1 | grep "^instance" | tee $O/touch.txt
