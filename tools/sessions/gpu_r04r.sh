#!/bin/bash
# round 4: rehearsal of `bench.py --gpus 2` at the driver's size (2 x 1.0 M tet10), both ranks on GPU 0 over gloo
set -o pipefail
O=gpurun_out/r04r; mkdir -p $O
FCVM_VERBOSE=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 800 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2x1M.json 2> $O/bench_2x1M.err; echo "rc=$?"
grep "smoother bounds" $O/bench_2x1M.json | head -2
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r04r/bench_2x1M.json').read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'pcg/solve', d['pcg_iterations_per_solve'], 'analysis', d['north_star_workload']['analysis_s'], 'LF', d['north_star_workload']['collapse_load_factor'], 'setup', d['setup_s'])
print(d['config']['preconditioner'][:200]); print(d['config']['partition'], d['config']['exchange'][-120:])
P
tail -3 $O/bench_2x1M.err | cut -c1-300
