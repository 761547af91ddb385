#!/bin/bash
# round 5: over-correction of the aggregate level (unsmoothed aggregation) with both fine smoothers, 1 M plate
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05x; mkdir -p $O
cd $R
for sc in 1.0 1.4 1.8; do
FCVM_MG_L2_SCALE=$sc timeout -k 10 900 python tools/sessions/patch_scan_r05.py plate:1.45 "FCVM_MG_PATCH=0" "FCVM_MG_PATCH=1" 2>&1 | grep -v "^fcvm" | sed "s/^/scale $sc: /" | tee -a $O/scan_over.txt
done
