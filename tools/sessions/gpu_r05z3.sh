#!/bin/bash
# round 5: shared corner nodes first in the restriction too: tests and the 2-rank rehearsal
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O; rm -f $O/restrict_split.txt
cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_configs.py -q -m gpu -k "boundary or rank or partition" 2>&1 | tail -4 | tee -a $O/restrict_split.txt
for sp in 1 0; do
FCVM_DIST_L1_SPLIT=$sp GPU_MAX_HW_QUEUES=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/bench2_rs$sp.json 2> $O/bench2_rs$sp.err; echo "rc=$?"
python3 - $O/bench2_rs$sp.json $sp <<'P' | tee -a $O/restrict_split.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("2 ranks, FCVM_DIST_L1_SPLIT=%s: pcg iterations per step %.2f, load factor %.12f, ms per step %.1f (one GPU shared, gloo)" % (sys.argv[2], d["pcg_iterations_per_step"], d["north_star_workload"]["collapse_load_factor"], d["ms_per_step"]))
P
done
