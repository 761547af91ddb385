#!/bin/bash
# the whole GPU suite + smoke, log under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/gpu_suite.log 2>&1
rc=$?
tail -15 gpurun_out/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
