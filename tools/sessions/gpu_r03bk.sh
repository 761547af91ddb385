#!/bin/bash
# round 3: what does the symmetric copy kernel cost with / without the 16-bit outputs, against the full-form copy
O=gpurun_out/r03bk
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in "FCVM_WIN_SYM_LO=0" "FCVM_WIN_SYM=0"; do
  env $v true
  export $v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o lo -- python3 $R/tools/lo_probe.py --reps 10 --solves 1 > $R/$O/lo.txt 2>&1 || exit 1
  python3 $R/tools/rocpd_top.py $R/$O/prof/lo_results.db 40 > $R/$O/lo_kernels.txt
  rm -f $R/$O/prof/*.db
  echo "== $v"; grep "repack" $R/$O/lo_kernels.txt | cut -c1-50,112-170
  unset ${v%%=*}
done
