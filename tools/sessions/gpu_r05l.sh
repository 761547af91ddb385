#!/bin/bash
# round 5: placement tuning of the SpMV's result vector / value array (tune_sym_placement): the vector-placement probe again, then the bench
set -o pipefail
O=gpurun_out/r05l; mkdir -p $O
FCVM_VERBOSE=1 timeout -k 10 400 python3 tools/sessions/spmv_vecplace_r05.py 2>&1 | grep -E "^instance . +(in PCG|the solver)|placement of" | cut -c1-330 | tee $O/vecplace_tuned.txt
FCVM_VERBOSE=1 timeout -k 10 300 python3 bench.py --no-cpu > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
grep "placement of" $O/bench.err
python3 - $O/bench.json <<'P'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'value', d['value'], 'pcg/solve', d['pcg_iterations_per_solve'], 'analysis', d['north_star_workload']['analysis_s'])
print('roofline', {k: d['roofline'][k] for k in ('achieved','frac','avg_launch_ms','back_to_back_launch_ms','stream_read_GBs_this_box')})
print('assembly', d['assembly']['ms'], 'setup', d['setup_s'])
print('gnly', {k: v for k, v in d['gnly'].items() if k != 'workload' and k != 'reference'})
P
