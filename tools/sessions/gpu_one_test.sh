#!/bin/bash
# one or a few GPU tests by -k expression: bash tools/sessions/gpu_one_test.sh "<expr>"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "$1" > gpurun_out/one_test.log 2>&1; rc=$?
tail -25 gpurun_out/one_test.log | cut -c1-220
exit $rc
