#!/bin/bash
O=gpurun_out/r02zd
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 700 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_configs.py -x -q --durations=6 -k "not config4 and not full_collapse" > $O/dist.log 2>&1; echo "rc $?"; tail -12 $O/dist.log | cut -c1-160
timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "fused_inner" > $O/mdot.log 2>&1; tail -3 $O/mdot.log | cut -c1-200
FCVM_DIST_SHARE_GPU=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --cells 30 > $O/bench2.log 2>&1; echo "bench2 rc $?"; grep -o '"ms_per_step": [0-9.]*\|"n_gpus": [0-9]*\|"pcg_iterations_per_step": [0-9.]*' $O/bench2.log
