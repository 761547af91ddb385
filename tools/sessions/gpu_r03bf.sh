#!/bin/bash
# round 3: Morton keys with one scale for the three axes (compact groups) against the per-axis scaling used so far -- kernel times
O=gpurun_out/r03bf
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
cd /tmp
for ax in 0 1; do
  export FCVM_MORTON_AXIS_SCALE=$ax
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_ax$ax -o lo -- python3 $R/tools/lo_probe.py --reps 60 --solves 3 > $R/$O/lo_ax$ax.txt 2>&1 || exit 1
  python3 $R/tools/rocpd_top.py $R/$O/prof_ax$ax/lo_results.db 12 > $R/$O/lo_ax${ax}_kernels.txt
  rm -f $R/$O/prof_ax$ax/*.db
  echo "== per-axis scaling $ax"; grep "cycle form\|solve:\|plain SpMV" $R/$O/lo_ax$ax.txt; grep "spmv_win\|l1_i16\|repack\|assemble_lds" $R/$O/lo_ax${ax}_kernels.txt | cut -c1-50,112-170
done
