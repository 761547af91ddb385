#!/bin/bash
# does the physical placement of the window value array explain the 0.465 / 0.513 ms SpMV between processes?
O=gpurun_out/r03ac
mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3 4; do
  FCVM_VERBOSE=1 FCVM_VALS_TRIES=4 timeout -k 10 200 python3 tools/lo_probe.py --reps 2 --solves 1 > $O/run$i.log 2>&1
  grep "candidate\|plain SpMV\|solve:" $O/run$i.log | tr '\n' ' '; echo
done
