#!/bin/bash
# round 4: geometric stiffness with rows in LDS -- its test and the buckling / GNLY tests
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "geometric or buckling or gnly or u100" > gpurun_out/r04w_tests.log 2>&1 || { tail -40 gpurun_out/r04w_tests.log; exit 1; }
tail -15 gpurun_out/r04w_tests.log
