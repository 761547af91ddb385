#!/bin/bash
# round 5: the exchange as one all_to_all_single: multi-rank GPU tests (gloo, ranks sharing the GPU), the one-rank RCCL run of the layer, the 2-rank rehearsal
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_configs.py -q -m gpu -k "rank or partition" 2>&1 | tail -3 | tee $O/a2a_tests.txt
FCVM_DIST_BACKEND=nccl FORCE_MODES=1 timeout -k 10 300 python3 tools/sessions/dist_host_overhead_r04.py 2>&1 | grep -E "host-sequenced|Error" | tail -2 | cut -c1-200 | tee -a $O/a2a_tests.txt
for ex in alltoall p2p; do
FCVM_DIST_EXCHANGE=$ex GPU_MAX_HW_QUEUES=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/bench2_$ex.json 2> $O/bench2_$ex.err; echo "rc=$?"
python3 - $O/bench2_$ex.json $ex <<'P' | tee -a $O/a2a_tests.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("2 ranks, FCVM_DIST_EXCHANGE=%s: pcg iterations per step %.2f, load factor %.9f, ms per step %.1f (one GPU shared, gloo)" % (sys.argv[2], d["pcg_iterations_per_step"], d["north_star_workload"]["collapse_load_factor"], d["ms_per_step"]))
P
done
