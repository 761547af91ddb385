#!/bin/bash
# round 5: the configs and multi-rank GPU tests with FCVM_POISON=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O
cd $R
FCVM_POISON=1 timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dist.py -q -m gpu > $O/configs_poison.log 2>&1
echo "rc=$? $(tail -1 $O/configs_poison.log)"
grep -n "^FAILED\|^E  " $O/configs_poison.log | cut -c1-250 | head -20
