#!/bin/bash
O=gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "16bit_smoother" > $O/tests.log 2>&1
echo tests rc=$?; tail -3 $O/tests.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py > $R/$O/lo_probe.log 2>&1; echo rc=$?
cd $R
grep "cycle form\|solve:" $O/lo_probe.log
python3 tools/rocpd_top.py $O/prof/lo_results.db 14 | cut -c1-60,105-160
rm -f $O/prof/*.db
