#!/bin/bash
# round 4: GNLY -- solutions of the previous tangent carried over as start directions (FCVM_GNLY_CARRY)
set -e
mkdir -p gpurun_out
for cdepth in 2 1 0 4; do
  FCVM_GNLY_CARRY=$cdepth timeout -k 10 300 python tools/gnly_probe.py --steps 8 > gpurun_out/r04zc_gnly_$cdepth.log 2>&1
  echo "carry $cdepth"; tail -5 gpurun_out/r04zc_gnly_$cdepth.log | cut -c1-170
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gnly or buckling or u100" > gpurun_out/r04zc_tests.log 2>&1 || { tail -40 gpurun_out/r04zc_tests.log; exit 1; }
tail -3 gpurun_out/r04zc_tests.log
