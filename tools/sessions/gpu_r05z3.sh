#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
TAG=alpha1s bash tools/sessions/gpu_r05x.sh "FCVM_MG_ALPHA1S=6" "FCVM_MG_ALPHA1S=9" "FCVM_MG_ALPHA1S=12" "FCVM_MG_ALPHA1S=18" "FCVM_MG_ALPHA1S=30" "FCVM_MG_ALPHA1S=60"
