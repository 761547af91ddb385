#!/bin/bash
# round 3: aggregate count of the distributed aggregate level, two ranks of 1.0 M tet10 on one GPU (iteration counts)
O=gpurun_out/r03bb
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
for a in 1700 2600; do
  FCVM_DIST_AGGREGATES=$a FCVM_DIST_SHARE_GPU=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 3 --warmup 2 > $O/bench2_$a.json 2> $O/bench2_$a.err || exit 1
  python3 -c "
import json
d=json.loads(open('$O/bench2_$a.json').read()); n=d['north_star_workload']
print($a, d['config']['preconditioner'][95:160], 'its/solve', round(n['pcg_iterations_per_solve'],1), 'analysis', round(n['analysis_s'],1), 'setup inside', round(n['setup_s_inside'],2), 'ms/step', round(d['ms_per_step'],1))"
done
