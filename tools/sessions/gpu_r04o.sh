#!/bin/bash
set -o pipefail
O=gpurun_out/r04o; mkdir -p $O
run() { # name, gpus, env...
  local name=$1; shift; local n=$1; shift
  env "$@" FCVM_VERBOSE=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus $n --plate-h 5 --steps 6 --warmup 3 > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"
  grep "smoother bounds" $O/$name.json $O/$name.err | head -2
  grep "RuntimeError" $O/$name.err | tail -1 | cut -c1-330; python3 -c "
import json,sys
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('  ms/step', d['ms_per_step'], 'pcg/solve', d['pcg_iterations_per_solve'], 'LF', d['north_star_workload']['collapse_load_factor'])
except Exception as e: print('  no result')
"
}
run r2_15 2 FCVM_X=0
run r2_60 2 FCVM_DIST_LMAX_ITERS=60
run r3_15 3 FCVM_X=0
run r3_60 3 FCVM_DIST_LMAX_ITERS=60
