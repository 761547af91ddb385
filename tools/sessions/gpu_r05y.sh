#!/bin/bash
# round 5: bench.py (plate analysis, cube, GNLY) with FCVM_POISON=1: same counts and load factor as without?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O
cd $R
FCVM_POISON=1 timeout -k 10 600 python3 bench.py --no-cpu > $O/bench_poison.json 2> $O/bench_poison.err; echo "rc=$?"
python3 - <<'P'
import json
d = json.loads(open("gpurun_out/r05y/bench_poison.json").read().strip().splitlines()[-1])
print("ms/step %.2f pcg/step %.2f analysis its/solve %.2f load factor %.12f cube its %.2f gnly its %.2f gnly ms %.2f" % (d["ms_per_step"], d["pcg_iterations_per_step"], d["north_star_workload"]["pcg_iterations_per_solve"], d["north_star_workload"]["collapse_load_factor"], d["configs2_cube"]["pcg_iterations_per_newton_iteration"], d["gnly"]["pcg_iterations_per_newton_iteration"], d["gnly"]["ms_per_newton_iteration"]))
P
