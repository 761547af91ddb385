#!/bin/bash
# round 5, last build: 6-rank rehearsal on one GPU (gloo) with the exchange as one collective and shared nodes first at every level
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
GPU_MAX_HW_QUEUES=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 6 --steps 20 --warmup 5 --no-cpu > $O/bench6.json 2> $O/bench6.err; echo "rc6=$?"
python3 - $O/bench6.json <<'P' | tee $O/bench6.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("6 ranks: pcg iterations per step %.2f, per solve of the analysis %.2f, load factor %.12f, ms per step %.1f (one GPU shared, gloo), device memory %s, aggregate level %s" % (d["pcg_iterations_per_step"], d["north_star_workload"]["pcg_iterations_per_solve"], d["north_star_workload"]["collapse_load_factor"], d["ms_per_step"], d["device_memory_GB"], d["aggregate_level"]))
P
