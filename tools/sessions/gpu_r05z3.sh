#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
MESH=emb:3 TAG=emb3 bash tools/sessions/gpu_r05x.sh "FCVM_MG_PATCH=0" "FCVM_MG_PATCH=1" "FCVM_MG_PATCH=1 FCVM_MG_K1S=6"
