#!/bin/bash
# the whole GPU suite twice in a row (looking for the one unexplained failure of the vertex-patch test), logs under gpurun_out/
mkdir -p gpurun_out
for k in 1 2; do
  timeout -k 10 560 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite_$k.log 2>&1
  echo "run $k: rc=$? $(tail -1 gpurun_out/gpu_suite_$k.log)"
done
