#!/bin/bash
O=gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "two_right_hand_sides" > $O/tests.log 2>&1
echo tests rc=$?; tail -25 $O/tests.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o g -- python3 $R/tools/gnly_probe.py > $R/$O/gnly_pair.log 2>&1; echo pair rc=$?; tail -3 $R/$O/gnly_pair.log
cd $R
python3 tools/rocpd_top.py $O/prof/g_results.db 16 | cut -c1-60,105-160
rm -f $O/prof/*.db
