#!/bin/bash
# round 3: aggregate-level product from the lower triangle of the inverse -- parity, kernel times, bench
O=gpurun_out/r03ap
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python3 -m pytest tests -q -m gpu -x -k "lower_triangle or sparse_aggregate or dense_spd or preconditioners or lockstep" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
cd /tmp
for sv in 1 0; do
  export FCVM_L2_SYMV=$sv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_sv$sv -o lo -- python3 $R/tools/lo_probe.py --reps 60 --solves 3 > $R/$O/lo_sv$sv.txt 2>&1 || exit 1
  python3 $R/tools/rocpd_top.py $R/$O/prof_sv$sv/lo_results.db 16 > $R/$O/lo_sv${sv}_kernels.txt
  rm -f $R/$O/prof_sv$sv/*.db
  grep "cycle form\|solve:" $R/$O/lo_sv$sv.txt; grep "gemv\|symv" $R/$O/lo_sv${sv}_kernels.txt | cut -c1-50,112-170
done
