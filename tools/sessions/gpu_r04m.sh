#!/bin/bash
set -o pipefail
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "straight_into or two_right or switches or window_format or bad_mesh or reruns" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
FCVM_PATTERN_TIMING=1 python tools/setup_profile.py --plate-h 1.45 2>&1 | grep "symmetric form\|GpuModel\|window format (rest\|first calc_gsm"
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r04m/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["assembly"], d["north_star_workload"]["analysis_s"], d["north_star_workload"]["setup_s_inside"], d["setup_s"], d["roofline"]["frac"])
P
