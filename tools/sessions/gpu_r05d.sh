#!/bin/bash
# round 5, verdict item 3: the FP64 SpMV on whatever box this lands on -- product build and the A/B builds of tools/sessions/build_ab_r05.sh
# (no transposed products / three workgroups per CU / plain LDS stores instead of 64-bit atomics / atomics without the fixed-point conversion),
# each in its own process, one after the other on the same box, with shader clock and power sampled from sysfs
set -o pipefail
O=gpurun_out/r05d; mkdir -p $O
for v in product notr minwaves3 plainlds nofixed product; do
  if [ $v != product ]; then
    lib=$GRAFT_REPO_ROOT/tools/sessions/ab/libfcvm_$v.so
    [ -f $lib ] || { echo "missing $lib: run tools/sessions/build_ab_r05.sh first"; exit 1; }
    export FCVM_HIP_LIB=$lib
  else
    unset FCVM_HIP_LIB
  fi
  AB_TAG=$v timeout -k 10 200 python3 tools/sessions/spmv_slow_box_r05.py 2>&1 | grep "^\[" | tee -a $O/slow_box.txt
done
