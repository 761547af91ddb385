#!/bin/bash
# round 5: first runs of the vertex-patch smoother: its test, then the bench with and without it
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05w; mkdir -p $O
cd $R
FCVM_VERBOSE=1 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "vertex_patch" 2>&1 | grep -v "^fcvm: window\|^fcvm: assembly\|^fcvm: symmetric" | tail -30 | tee $O/test.txt
for p in 1 0; do
  echo "== FCVM_MG_PATCH=$p" | tee -a $O/bench.txt
  FCVM_MG_PATCH=$p FCVM_VERBOSE=1 timeout -k 10 400 python bench.py --no-cpu --no-cube --no-gnly > $O/bench_$p.json 2> $O/bench_$p.err && python - $O/bench_$p.json <<'PY' | tee -a $O/bench.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step %.2f  pcg/step %.2f  analysis_s %.3f  its/solve(analysis) %.2f  lf %.9f  avg_spmv_ms %.4f" % (d["ms_per_step"], d["pcg_iterations_per_step"], d["north_star_workload"]["analysis_s"], d["north_star_workload"]["pcg_iterations_per_solve"], d["north_star_workload"]["collapse_load_factor"], d["roofline"]["avg_launch_ms"]))
PY
  grep "vertex-patch" $O/bench_$p.err | head -5 | tee -a $O/bench.txt
done
