#!/bin/bash
# round 5, the search for the solve that broke down only late in whole-suite runs and the audit that followed (profiles/r05_poison_audit.txt); the steps were
# run in separate calls.  STEP selects one: stress | stress2 | suitediag | probe | poison_parity | poison_configs | poison_bench | smallstream | all
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O
cd $R
S=${STEP:-all}
want() { [ "$S" = all ] || [ "$S" = "$1" ]; }
if want stress; then timeout -k 10 600 python tools/sessions/patch_stress_r05.py 25 24 2>&1 | grep -v "^fcvm" | tee $O/stress.txt; fi
if want stress2; then timeout -k 10 900 python tools/sessions/patch_stress2_r05.py 30 > $O/stress2.txt 2>&1; grep -E "FAILED|failures" $O/stress2.txt | tail -3; fi
if want suitediag; then timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/suite_diag.log 2>&1; echo "rc=$? $(tail -1 $O/suite_diag.log)"; grep -n "before the solve\|leg:\|not finite\|maxit" $O/suite_diag.log | cut -c1-600; fi
if want probe; then for mesh in cube:30 cube:60 emb:1 plate:4; do echo "== $mesh"; timeout -k 10 300 python tools/sessions/poison_probe_r05.py $mesh 2>&1 | grep -v "^fcvm\|finite$" | cut -c1-300; done | tee $O/probe.txt; fi
if want poison_parity; then FCVM_POISON=1 timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -q -m gpu > $O/parity_poison.log 2>&1; echo "rc=$? $(tail -1 $O/parity_poison.log)"; fi
if want poison_configs; then FCVM_POISON=1 timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dist.py -q -m gpu > $O/configs_poison.log 2>&1; echo "rc=$? $(tail -1 $O/configs_poison.log)"; fi
if want poison_bench; then FCVM_POISON=1 timeout -k 10 600 python3 bench.py --no-cpu > $O/bench_poison.json 2> $O/bench_poison.err; echo "rc=$?"; tail -c 400 $O/bench_poison.json; fi
if want smallstream; then timeout -k 10 120 python tools/sessions/small_stream_r05.py 2>&1 | tee $O/small_stream.txt; fi
