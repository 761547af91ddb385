#!/bin/bash
# round 5: the replicated aggregate level of the multi-GPU cycle (dense / sparse + super-aggregates) -- GPU tests of tests/test_gpu_dist.py,
# then the rehearsal of `bench.py --gpus 2` with both ranks on GPU 0 over gloo at 2 x 0.25 M and 2 x 1.0 M tet10 (round 4: 47.8 PCG
# iterations per solve with 2108 dense aggregates)
set -o pipefail
O=gpurun_out/r05a; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu -s > $O/gpu_dist.log 2>&1; rc=$?
grep -E "three ranks|passed|failed|Error" $O/gpu_dist.log | cut -c1-400
[ $rc -ne 0 ] && { tail -30 $O/gpu_dist.log; exit 1; }
for h in small full; do
  if [ $h = small ]; then ph=2.9; else ph=1.45; fi
  FCVM_VERBOSE=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 5 --plate-h $ph > $O/bench_$h.json 2> $O/bench_$h.err; rc=$?; echo "rc=$rc"
  [ $rc -ne 0 ] && { tail -20 $O/bench_$h.err | cut -c1-400; exit 1; }
  python3 - $O/bench_$h.json <<'P'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'pcg/solve', d.get('pcg_iterations_per_solve'), 'analysis', d['north_star_workload']['analysis_s'], 'LF', d['north_star_workload']['collapse_load_factor'])
print(d['config']['workload'][:160]); print(d['config'].get('exchange'))
P
  grep -E "fcvm dist|sparse aggregate" $O/bench_$h.err | cut -c1-400
done
