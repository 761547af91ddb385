#!/bin/bash
# round 3: 16-bit smoother passes on the symmetric form too -- parity, then kernel times
O=gpurun_out/r03bi
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 500 python3 -m pytest tests -q -m gpu -x -k "symmetric_window or window_format_spmv or lockstep or reruns or 16bit or preconditioners" > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -25 $O/tests.log | cut -c1-220
[ $rc -eq 0 ] || exit 1
cd /tmp
FCVM_VERBOSE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py --reps 60 --solves 3 > $R/$O/lo.txt 2>&1 || exit 1
python3 $R/tools/rocpd_top.py $R/$O/prof/lo_results.db 14 > $R/$O/lo_kernels.txt
rm -f $R/$O/prof/*.db
grep "cycle form\|solve:\|plain SpMV\|symmetric window" $R/$O/lo.txt; cut -c1-50,112-170 $R/$O/lo_kernels.txt | head -10
