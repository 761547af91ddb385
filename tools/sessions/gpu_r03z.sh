#!/bin/bash
O=gpurun_out/r03z
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/prof -o g -- python3 $R/tools/gnly_probe.py --steps 4 > $R/$O/gnly.log 2>&1; echo rc=$?
cd $R
grep iteration $O/gnly.log | tail -2
python3 tools/rocpd_timeline.py $O/prof/g_results.db k_assemble_lds 3 2>/dev/null | head -24
rm -f $O/prof/*.db
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q > $O/tests.log 2>&1; echo rc=$?; tail -4 $O/tests.log
