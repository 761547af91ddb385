#!/bin/bash
# round 3: tangent assembly with the plastic part as a rank-one update -- parity tests, GNLY iteration cost, kernel trace
O=gpurun_out/r03af
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 500 python3 -m pytest tests -q -m gpu -k "tangent or gnly or assemble or dist or buckling" > $O/tests.log 2>&1; echo tests rc=$?; tail -5 $O/tests.log
timeout -k 10 200 python3 tools/gnly_probe.py --cells 55 > $O/gnly.txt 2>&1; echo gnly rc=$?; tail -8 $O/gnly.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o gnly -- python3 $R/tools/gnly_probe.py --cells 55 > $R/$O/gnly_prof.txt 2>&1; echo prof rc=$?
cd $R
python3 tools/rocpd_top.py $O/prof/gnly_results.db 25 > $O/gnly_kernel_stats.txt; head -30 $O/gnly_kernel_stats.txt
rm -f $O/prof/*.db
