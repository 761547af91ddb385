#!/bin/bash
# round 5: small aggregates on the sparse aggregate level (now 2 Chebyshev steps per sweep) against the dense level of 72-node aggregates, 1 M plate, block-Jacobi fine smoother
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05x; mkdir -p $O
cd $R
timeout -k 10 1000 python tools/sessions/patch_scan_r05.py plate:1.45 "FCVM_MG_PATCH=0" "FCVM_MG_TARGET=48" "FCVM_MG_TARGET=32" "FCVM_MG_TARGET=24" "FCVM_MG_TARGET=16" "FCVM_MG_TARGET=24 FCVM_MG_K1S=2" "FCVM_MG_TARGET=24 FCVM_L2S_K=3" "FCVM_MG_TARGET=24 FCVM_L2S_COARSE=1" "FCVM_MG_TARGET=16 FCVM_MG_K1S=2" 2>&1 | grep -v "^fcvm" | tee $O/scan_small_aggregates.txt
