#!/bin/bash
# round 5: the sparse aggregate level with 2 smoothing steps per sweep -- the 6-rank rehearsal again (default now 2 / 2 for the replicated level), and the
# one-GPU solver on the 6.0 M plate with FCVM_L2S_K=2 against its default 8 (does the one-GPU default want the same change?)
set -o pipefail
O=gpurun_out/r05s; mkdir -p $O
FCVM_VERBOSE=1 FCVM_SETUP_TIMING=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 800 python3 bench.py --gpus 6 --steps 20 --warmup 5 --no-cpu > $O/bench6.json 2> $O/bench6.err; echo "rc=$?"
python3 - $O/bench6.json <<'P'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('6 ranks: pcg/solve', d.get('pcg_iterations_per_solve'), 'analysis pcg/solve', d['north_star_workload']['pcg_iterations_per_solve'], 'LF', d['north_star_workload']['collapse_load_factor'], d['aggregate_level'])
P
for k in 2 8; do
FCVM_L2S_K=$k timeout -k 10 500 python3 bench.py --plate-h 0.79796 --no-cpu --no-cube > $O/bench1_k$k.json 2> $O/bench1_k$k.err; echo "one GPU K=$k rc=$?"
python3 - $O/bench1_k$k.json <<'P'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('  ms/step', d['ms_per_step'], 'pcg/solve', d['pcg_iterations_per_solve'], 'analysis', d['north_star_workload']['analysis_s'], 'pcg/solve analysis', d['north_star_workload']['pcg_iterations_per_solve'])
P
done
