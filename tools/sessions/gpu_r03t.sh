#!/bin/bash
# round 3, call t: rehearsal of the N > 1 bench on the plate workload, 2 and 3 ranks sharing GPU 0 over gloo
O=gpurun_out/r03t
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
for n in 2 3; do
  FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus $n --plate-h 6 --steps 6 --warmup 3 > $O/bench_${n}ranks.json 2> $O/bench_${n}ranks.err; echo dist bench $n rc=$?
  grep -v "socket.cpp\|amdgpu.ids\|Gloo" $O/bench_${n}ranks.err | tail -8; head -c 2500 $O/bench_${n}ranks.json; echo
done
timeout -k 10 200 python3 bench.py --plate-h 6 --steps 6 --warmup 3 --no-cpu --no-cube > $O/bench_1rank.json 2> $O/bench_1rank.err; echo one rank rc=$?; head -c 600 $O/bench_1rank.json
