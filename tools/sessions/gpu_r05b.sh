#!/bin/bash
# round 5: the whole GPU suite on the build with the replicated aggregate level / grading guard, then the default bench (with the gnly object)
set -o pipefail
O=gpurun_out/r05b; mkdir -p $O
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; rc=$?
tail -5 $O/gpu_suite.log | cut -c1-300
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $O/gpu_suite.log | head -30 | cut -c1-300; exit 1; }
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench rc=$rc"
[ $rc -ne 0 ] && { tail -20 $O/bench_default.err | cut -c1-300; exit 1; }
python3 - $O/bench_default.json <<'P'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'value', d['value'], 'pcg/solve', d['pcg_iterations_per_solve'], 'analysis', d['north_star_workload']['analysis_s'])
print('roofline', {k: d['roofline'][k] for k in ('achieved','frac','avg_launch_ms','back_to_back_launch_ms','stream_read_GBs_this_box')})
print('assembly', d['assembly']['ms'], 'setup', d['setup_s'])
print('gnly', d.get('gnly'))
print('cube', d.get('configs2_cube'))
P
