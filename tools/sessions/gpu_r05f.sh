#!/bin/bash
# round 5, verdict item 3: FP64 SpMV / 16-bit pass / stream read against load history and the card's sensors (thermal or power state?)
set -o pipefail
O=gpurun_out/r05f; mkdir -p $O
timeout -k 10 300 python3 tools/sessions/spmv_thermal_r05.py 2>&1 | grep -v "^$" | tee $O/thermal.txt | cut -c1-420
