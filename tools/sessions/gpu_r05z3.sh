#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/sessions/dist_host_profile_r05.py 2>&1 | grep -v "^fcvm" | tail -45 | cut -c1-200 | tee $O/host_profile.txt
