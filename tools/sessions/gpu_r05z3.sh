#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
timeout -k 10 200 python3 tools/sessions/l1_pass_parts_r05.py 2>&1 | grep -v "^fcvm" | tail -3 | tee $O/l1_parts.txt
