#!/bin/bash
# round 4, third GPU call: assembly straight into the symmetric window form -- parity test first, then the bench
set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "straight_into" > $O/tests_direct.log 2>&1; rc=$?; echo "rc=$rc" >> $O/tests_direct.log; tail -25 $O/tests_direct.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
FCVM_ASSEMBLE_SYM=0 timeout -k 10 600 python bench.py > $O/bench_copy.json 2> $O/bench_copy.err
python - <<'P'
import json
for f in ("bench", "bench_copy"):
    d = json.loads(open("gpurun_out/r04c/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["assembly"], d["north_star_workload"]["analysis_s"], d["north_star_workload"]["collapse_load_factor"], d["setup_s"])
P
