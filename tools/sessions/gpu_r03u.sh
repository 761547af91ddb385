#!/bin/bash
# round 3, call u: distributed multiplicative cycle (16-bit smoother products per rank) on the GPU: dist tests, plate bench rehearsal
O=gpurun_out/r03u
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 700 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_configs.py -x -q --durations=6 -k "not cube_8m" > $O/dist_tests.log 2>&1; echo tests rc=$?; tail -14 $O/dist_tests.log
for n in 2 3; do
  FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus $n --plate-h 6 --steps 6 --warmup 3 > $O/bench_${n}ranks.json 2> $O/bench_${n}ranks.err; echo dist bench $n rc=$?
  python3 -c "
import json; d=json.load(open('$O/bench_${n}ranks.json')); print(d['ms_per_step'], d['pcg_iterations_per_solve'], d['config']['exchange'][-60:], d['north_star_workload']['collapse_load_factor'], d['north_star_workload']['analysis_s'])"
done
FCVM_DIST_CYCLE=additive FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --plate-h 6 --steps 6 --warmup 3 > $O/bench_2ranks_additive.json 2> $O/bench_2ranks_additive.err; echo additive rc=$?
python3 -c "
import json; d=json.load(open('$O/bench_2ranks_additive.json')); print(d['ms_per_step'], d['pcg_iterations_per_solve'], d['config']['exchange'][-60:], d['north_star_workload']['collapse_load_factor'], d['north_star_workload']['analysis_s'])"
