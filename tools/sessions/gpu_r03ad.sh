#!/bin/bash
# where does the time of k_spmv_win_lo go: variants without the x staging / without the matrix loop (results are garbage: timing only)
O=gpurun_out/r03ad
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in "" _skipSTAGE _skipLOOP; do
  FCVM_HIP_LIB=$R/fcvm-workbench_amd/fcvm_hip/libfcvm_hip$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof$v -o lo -- python3 $R/tools/lo_probe.py --reps 20 --solves 0 > $R/$O/lo_probe$v.log 2>&1
  echo "== variant '$v'"; python3 $R/tools/rocpd_top.py $R/$O/prof$v/lo_results.db 12 | grep "win_lo" | cut -c1-60,105-160
  rm -f $R/$O/prof$v/*.db
done
