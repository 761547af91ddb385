#!/bin/bash
# round 3: prolongation fused into the staging of the post-smoothing pass -- parity (bitwise against the pair path), A/B on one box
O=gpurun_out/r03bt
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 500 python3 -m pytest tests -q -m gpu -x -k "symmetric_window or lockstep or reruns or 16bit or preconditioners" > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -4 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
cd /tmp
for v in 1 0 1 0; do
  export FCVM_MG_FUSE_PROLONG=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py --reps 400 --solves 6 > $R/$O/lo_$v.txt 2>&1 || exit 1
  python3 $R/tools/rocpd_top.py $R/$O/prof/lo_results.db 30 > $R/$O/k.txt; rm -f $R/$O/prof/*.db
  echo "== fused $v"; grep "cycle form" $R/$O/lo_$v.txt; grep "solve:" $R/$O/lo_$v.txt | tail -2; grep "lo_sym<2\|prolong_add" $R/$O/k.txt | cut -c1-50,112-170
done
