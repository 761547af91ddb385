#!/bin/bash
# round 3, refreshed after the aggregate-size change: CPU (oracle port, SuperLU) against GPU Newton iterations/s on one mesh per row, three sizes
O=gpurun_out/r03aw
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1150 python3 tools/cpu_gpu_table.py --h 6.0,4.0,3.0 --out $O/cpu_gpu_table.json > $O/table.log 2>&1
echo rc=$?; tail -c 1500 $O/table.log
