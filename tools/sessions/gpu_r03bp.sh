#!/bin/bash
# round 3: A/B of two builds (default vs fcvm_hip/libfcvm_hip_ab.so), one box, long runs; "f64" = the alternative build
O=gpurun_out/r03bp
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
cd /tmp
for v in int f64 int f64; do  # (f64 = the alternative build under test)
  if [ $v = f64 ]; then export FCVM_HIP_LIB=$R/fcvm-workbench_amd/fcvm_hip/libfcvm_hip_ab.so; else unset FCVM_HIP_LIB; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py --reps 400 --solves 6 > $R/$O/lo_$v.txt 2>&1 || exit 1
  python3 $R/tools/rocpd_top.py $R/$O/prof/lo_results.db 8 > $R/$O/k.txt; rm -f $R/$O/prof/*.db
  echo "== $v"; grep "spmv_win" $R/$O/k.txt | cut -c1-50,112-170
done
