#!/bin/bash
set -o pipefail
O=gpurun_out/r04u; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r04u/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["pcg_iterations_per_solve"], d["assembly"], d["north_star_workload"]["analysis_s"], d["north_star_workload"]["setup_s_inside"], d["setup_s"], d["roofline"]["frac"])
P
python tools/gnly_probe.py --cells 55 --steps 4 2>&1 | grep "iteration [23]"
