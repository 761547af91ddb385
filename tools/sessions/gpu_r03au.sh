#!/bin/bash
# round 3: level-1 SpMV with the epilogue operands loaded before the record loop -- kernel time, bitwise tests
O=gpurun_out/r03au
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests -q -m gpu -x -k "reruns or lockstep or preconditioners or 16bit or window_format" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py --reps 60 --solves 3 > $R/$O/lo.txt 2>&1 || exit 1
python3 $R/tools/rocpd_top.py $R/$O/prof/lo_results.db 16 > $R/$O/lo_kernels.txt
rm -f $R/$O/prof/*.db
grep "cycle form\|solve:" $R/$O/lo.txt; cut -c1-50,112-170 $R/$O/lo_kernels.txt
