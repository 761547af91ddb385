#!/bin/bash
# round 3: level-1 16-bit SpMV with plain vs nontemporal loads (does the matrix stay in the memory-side cache between the passes
# of one V-cycle?), and what the 0.42 s of set-up inside the analysis consists of
O=gpurun_out/r03ag
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
cd /tmp
for nt in 1 0; do
  export FCVM_L1_NT=$nt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_nt$nt -o lo -- python3 $R/tools/lo_probe.py --reps 60 --solves 3 > $R/$O/lo_nt$nt.txt 2>&1 || exit 1
  python3 $R/tools/rocpd_top.py $R/$O/prof_nt$nt/lo_results.db 14 > $R/$O/lo_nt${nt}_kernels.txt
  rm -f $R/$O/prof_nt$nt/*.db
  grep "cycle form\|solve:" $R/$O/lo_nt$nt.txt; grep "l1_i16\|spmv_win_lo\|gemv\|restrict\|prolong" $R/$O/lo_nt${nt}_kernels.txt | cut -c1-50,112-170
done
unset FCVM_L1_NT
cd $R
FCVM_PATTERN_TIMING=1 timeout -k 10 200 python3 tools/setup_inside.py > $O/setup_inside.txt 2>&1; grep -v "^ *$" $O/setup_inside.txt | head -60
