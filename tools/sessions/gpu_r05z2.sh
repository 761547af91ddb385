#!/bin/bash
# round 5: two level-1 Chebyshev steps per side in the multi-GPU cycle (FCVM_DIST_K1S=2: six exchange points per PCG iteration instead of eight): iterations, 2 ranks on one GPU over gloo
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z2; mkdir -p $O
cd $R
for k in 3 2; do
  FCVM_DIST_K1S=$k GPU_MAX_HW_QUEUES=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/bench2_k$k.json 2> $O/bench2_k$k.err; echo "rc=$?"
  python3 - $O/bench2_k$k.json $k <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("FCVM_DIST_K1S=%s: pcg iterations per step %.2f, per solve of the analysis %.2f, load factor %.9f, ms per step %.1f (one GPU shared, gloo)" % (sys.argv[2], d["pcg_iterations_per_step"], d["north_star_workload"]["pcg_iterations_per_solve"], d["north_star_workload"]["collapse_load_factor"], d["ms_per_step"]))
P
  grep -o "exchanges[^,]*,[^,]*" $O/bench2_k$k.err | tail -1
done
