#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O
cd $R
timeout -k 10 100 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "dense_inverse_product or vertex_patch" 2>&1 | tail -5
