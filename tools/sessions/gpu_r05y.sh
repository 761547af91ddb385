#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -q -m gpu -k "fp32_twin or rank" 2>&1 | tail -5
