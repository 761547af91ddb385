#!/bin/bash
# round 5: runner of tools/sessions/patch_scan_r05.py (one elastic solve per setting; iterations and ms per iteration).  The scans quoted in
# profiles/r05_patch_smoother.txt were taken with it:  MESH=plate:1.45 bash gpu_r05x.sh "FCVM_MG_PATCH=0" "FCVM_MG_PATCH=1 FCVM_MG_K1S=6" ...
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05x; mkdir -p $O
cd $R
timeout -k 10 1000 python tools/sessions/patch_scan_r05.py ${MESH:-plate:1.45} "$@" 2>&1 | grep -v "^fcvm" | tee -a $O/scan_${TAG:-more}.txt
