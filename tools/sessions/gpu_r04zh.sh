#!/bin/bash
# round 4 (late): rehearsal of `bench.py --gpus 4` (all ranks on GPU 0 over gloo, 4 x 46 k tet10) after the merged level-1 exchange
set -o pipefail
O=gpurun_out/r04zh; mkdir -p $O
FCVM_DIST_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 4 --steps 20 --warmup 5 --plate-h 4.0 > $O/bench4.json 2> $O/bench4.err; echo "rc=$?"
python3 - $O/bench4.json <<'P'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'pcg/solve', d.get('pcg_iterations_per_solve'), 'analysis', d['north_star_workload']['analysis_s'], 'LF', d['north_star_workload']['collapse_load_factor'])
print(d['config']['workload'][:160]); print(d['config'].get('partition')); print({k: d[k] for k in d if 'comm' in k or 'exchange' in k})
P
tail -2 $O/bench4.err | cut -c1-300
