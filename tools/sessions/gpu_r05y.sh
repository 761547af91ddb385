#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O
cd $R
timeout -k 10 900 python tools/sessions/patch_stress2_r05.py 30 > $O/stress2.txt 2>&1
grep -E "FAILED|failures|broke|largest eigen" $O/stress2.txt | sort | uniq -c | sort -rn | head -20
true
