#!/bin/bash
O=gpurun_out/r03ae
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 700 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_configs.py -x -q --durations=5 -k "not cube_8m" > $O/dist_tests.log 2>&1; echo tests rc=$?; tail -10 $O/dist_tests.log
for n in 2 3; do
  FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus $n --plate-h 6 --steps 6 --warmup 3 > $O/bench_${n}ranks.json 2> $O/bench_${n}ranks.err; echo dist bench $n rc=$?
  python3 - <<PY
import json
t=open('$O/bench_${n}ranks.json').read(); d=json.loads(t[t.index('{"metric'):].splitlines()[0])
print(round(d['ms_per_step'],1), d['pcg_iterations_per_solve'], d['north_star_workload']['collapse_load_factor'], round(d['north_star_workload']['analysis_s'],2))
PY
done
