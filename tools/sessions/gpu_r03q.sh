#!/bin/bash
O=gpurun_out/r03q
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 python3 tools/pair_debug.py emb > $O/pair_emb.log 2>&1; echo rc=$?; cat $O/pair_emb.log | tail -12
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python3 $R/tools/pair_debug.py plate time > $R/$O/pair_plate.log 2>&1; echo rc=$?; tail -14 $R/$O/pair_plate.log | grep -v "^W2026"
cd $R
python3 tools/rocpd_top.py $O/prof/p_results.db 12 | cut -c1-60,105-160
rm -f $O/prof/*.db
