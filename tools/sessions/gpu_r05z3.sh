#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
timeout -k 10 120 python3 tools/sessions/rccl_a2a_zero_r05.py 2>&1 | tail -3
