#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
TAG=l1_wpb bash tools/sessions/gpu_r05x.sh "FCVM_L1_WPB=1" "FCVM_L1_WPB=2" "FCVM_L1_WPB=4" "FCVM_L1_WPB=1" "FCVM_L1_WPB=2" "FCVM_L1_WPB=4"
