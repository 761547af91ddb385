#!/bin/bash
# round 5, verdict item 3: group -> workgroup maps of the symmetric-form kernels against the placement of a process's arrays
set -o pipefail
O=gpurun_out/r05g; mkdir -p $O
timeout -k 10 500 python3 tools/sessions/gmap_probe_r05.py 2>&1 | grep -E "^instance|^stream" | tee $O/gmap.txt
