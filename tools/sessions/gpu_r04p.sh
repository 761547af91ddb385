#!/bin/bash
set -o pipefail
O=gpurun_out/r04p; mkdir -p $O
python tools/sessions/lmax_scan_r04.py plate 2>&1 | grep "power iter" | head -2
python tools/sessions/lmax_scan_r04.py embankment 2>&1 | grep "power iter" | head -2
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
run r2 2 FCVM_X=0
run r3 3 FCVM_X=0
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r04p/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["pcg_iterations_per_solve"], d["assembly"]["ms"], d["north_star_workload"]["analysis_s"], d["north_star_workload"]["pcg_iterations_per_solve"], d["north_star_workload"]["setup_s_inside"], d["setup_s"], d["roofline"]["frac"])
P
