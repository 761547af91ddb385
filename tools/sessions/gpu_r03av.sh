#!/bin/bash
# round 3: slot reduction of the symmetric aggregate-level product -- parity and kernel time
O=gpurun_out/r03av
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests -q -m gpu -x -k "lower_triangle or reruns or lockstep or preconditioners" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py --reps 60 --solves 3 > $R/$O/lo.txt 2>&1 || exit 1
python3 $R/tools/rocpd_top.py $R/$O/prof/lo_results.db 30 > $R/$O/lo_kernels.txt
rm -f $R/$O/prof/*.db
grep "cycle form\|solve:" $R/$O/lo.txt; grep "symv\|l2_\|restrict\|prolong" $R/$O/lo_kernels.txt | cut -c1-50,112-170
