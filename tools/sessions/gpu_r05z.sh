#!/bin/bash
# round 5, final build: 2-rank and 6-rank rehearsal of bench.py --gpus N on one GPU (gloo; counts, memory and set-up only) after the late changes to the sparse aggregate level
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O
cd $R
GPU_MAX_HW_QUEUES=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/bench2.json 2> $O/bench2.err; echo "rc2=$?"
tail -c 1500 $O/bench2.json
GPU_MAX_HW_QUEUES=1 FCVM_VERBOSE=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 6 --steps 20 --warmup 5 --no-cpu > $O/bench6.json 2> $O/bench6.err; echo "rc6=$?"
tail -c 1500 $O/bench6.json
