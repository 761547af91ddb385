#!/bin/bash
# round 3: two ranks on one GPU through bench.py (rehearsal of the N > 1 launch path after the set-up changes), and the
# distributed GPU tests
O=gpurun_out/r03al
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
FCVM_DIST_SHARE_GPU=1 timeout -k 10 500 python3 bench.py --gpus 2 --plate-h 6 --steps 6 --warmup 3 > $O/bench2.json 2> $O/bench2.err; echo bench2 rc=$?; tail -c 1500 $O/bench2.json
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py -q > $O/dist_tests.log 2>&1; echo dist rc=$?; tail -3 $O/dist_tests.log
