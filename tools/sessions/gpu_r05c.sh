#!/bin/bash
# round 5: dress rehearsal of `bench.py --gpus 6` at the driver's per-rank size (6 x 1.0 M tet10 of the plate), all ranks on GPU 0 over gloo
# (a GPU box admits at most six processes on its card: the 8-rank case cannot be rehearsed), then the SAME mesh on one GPU for the
# PCG iteration count of the one-GPU solver
set -o pipefail
O=gpurun_out/r05c; mkdir -p $O
t0=$(date +%s)
FCVM_VERBOSE=1 FCVM_SETUP_TIMING=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 800 python3 bench.py --gpus 6 --steps 20 --warmup 5 > $O/bench6.json 2> $O/bench6.err; rc=$?
t1=$(date +%s); echo "rc=$rc wall $((t1-t0)) s"
grep -E "fcvm dist|sparse aggregate" $O/bench6.err | sort | uniq -c | cut -c1-700
[ $rc -ne 0 ] && { tail -30 $O/bench6.err | cut -c1-400; exit 1; }
python3 - $O/bench6.json <<'P'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'pcg/solve', d.get('pcg_iterations_per_solve'), 'analysis', d['north_star_workload']['analysis_s'], 'LF', d['north_star_workload']['collapse_load_factor'], 'pcg/solve analysis', d['north_star_workload']['pcg_iterations_per_solve'])
print(d['config']['workload'][:200]); print(d['config']['exchange']); print(d['config']['preconditioner'])
print('setup', d['setup_s']); print('mg setup', d['multigrid_setup_s']); print('memory', d['device_memory_GB']); print('l2', d['aggregate_level'])
print('roofline', d['roofline']); print('cpu_baseline' in d, d['config']['partition'])
P
timeout -k 10 500 python3 bench.py --plate-h 0.79796 --no-cpu --no-cube > $O/bench1_same_mesh.json 2> $O/bench1_same_mesh.err; rc=$?; echo "one GPU rc=$rc"
[ $rc -ne 0 ] && { tail -20 $O/bench1_same_mesh.err | cut -c1-300; exit 1; }
python3 - $O/bench1_same_mesh.json <<'P'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('one GPU, same mesh: ms/step', d['ms_per_step'], 'pcg/solve', d['pcg_iterations_per_solve'], 'analysis', d['north_star_workload']['analysis_s'], 'LF', d['north_star_workload']['collapse_load_factor'], 'pcg/solve analysis', d['north_star_workload']['pcg_iterations_per_solve'])
print(d['config']['workload'][:120]); print(d['config']['preconditioner'])
P
