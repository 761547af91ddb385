#!/bin/bash
# round 3: A/B of the block order inside the rows of the symmetric form (transposed-product blocks first vs column order), one box
O=gpurun_out/r03bs
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
cd /tmp
for v in 1 0 1 0; do
  export FCVM_SYM_FLAGGED_FIRST=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py --reps 400 --solves 6 > $R/$O/lo_$v.txt 2>&1 || exit 1
  python3 $R/tools/rocpd_top.py $R/$O/prof/lo_results.db 8 > $R/$O/k.txt; rm -f $R/$O/prof/*.db
  echo "== flagged first $v"; grep "spmv_win" $R/$O/k.txt | cut -c1-50,112-170
done
