#!/bin/bash
# round 4: fused end of the distributed cycle -- GPU dist tests, host-sequenced time per iteration with and without
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r04zb_tests.log 2>&1 || { tail -60 gpurun_out/r04zb_tests.log; exit 1; }
tail -3 gpurun_out/r04zb_tests.log
FORCE_MODES=1 timeout -k 10 300 python3 tools/sessions/dist_host_overhead_r04.py > gpurun_out/r04zb_on.log 2>&1
FCVM_DIST_FUSED_LO=0 FORCE_MODES=1 timeout -k 10 300 python3 tools/sessions/dist_host_overhead_r04.py > gpurun_out/r04zb_off.log 2>&1
grep host-sequenced gpurun_out/r04zb_on.log; grep host-sequenced gpurun_out/r04zb_off.log
