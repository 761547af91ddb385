#!/bin/bash
set -o pipefail
O=gpurun_out/r04v; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "switches or tangent or both_preconditioners or sparse_aggregate" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || { tail -40 $O/tests.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  FCVM_TET4_LDS=$v ROUTES=direct rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$v -o a -- python3 $R/tools/sessions/asm_time_r04.py tangent > $R/$O/asm$v.log 2>&1
  grep calc_gsm $R/$O/asm$v.log | sed "s/^/TET4_LDS=$v /" | cut -c1-120
  f=$(find $R/$O/prof$v -name "*kernel_stats.csv" | head -1)
  grep "tet4" "$f" | cut -c1-60,200-400
  rm -rf $R/$O/prof$v
done
