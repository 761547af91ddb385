#!/bin/bash
# round 4 (late): rehearsal of `bench.py --gpus 2` after the fused multi-GPU cycle, both ranks on GPU 0 over gloo, 2 x 0.25 M and 2 x 1.0 M tet10
set -o pipefail
O=gpurun_out/r04zg; mkdir -p $O
for h in small full; do
  if [ $h = small ]; then ph=2.9; else ph=1.45; fi
  FCVM_DIST_SHARE_GPU=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 5 --plate-h $ph > $O/bench_$h.json 2> $O/bench_$h.err; echo "rc=$?"
  python3 - $O/bench_$h.json <<'P'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'pcg/solve', d.get('pcg_iterations_per_solve'), 'analysis', d['north_star_workload']['analysis_s'], 'LF', d['north_star_workload']['collapse_load_factor'])
print(d['config']['workload'][:160]); print(d.get('comm_per_pcg_iteration'), d['config'].get('partition'))
P
  tail -2 $O/bench_$h.err | cut -c1-300
done
