#!/bin/bash
# round 5, the late multi-GPU work: the commands behind profiles/r05_rccl_one_rank.txt and the later sections of profiles/r05_dist_rehearsal_final_build.txt
# (they were run in several calls; every step is independent).  STEP selects one: rccl | p2p | hostprofile | tests | rehearsal2 | rehearsal6 | l1parts | all
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
S=${STEP:-all}
want() { [ "$S" = all ] || [ "$S" = "$1" ]; }
if want rccl; then       # the layer's collectives over RCCL on one rank (backend nccl, world size 1) against gloo
  for b in gloo nccl; do echo "== backend $b"; FCVM_DIST_BACKEND=$b FORCE_MODES=1 timeout -k 10 300 python3 tools/sessions/dist_host_overhead_r04.py 2>&1 | grep -E "host-sequenced|Error|RCCL" | tail -4 | cut -c1-300; done | tee $O/rccl_one_rank.txt
fi
if want p2p; then        # point-to-point as the Exchanger issues it, self as peer; host cost split; all_to_all_single; all-zero splits
  timeout -k 10 120 python3 tools/sessions/rccl_p2p_self_r05.py 2>&1 | grep -E "rows|host time|all_to_all" | tee $O/rccl_p2p_self.txt
  timeout -k 10 120 python3 tools/sessions/rccl_a2a_zero_r05.py 2>&1 | tail -1
fi
if want hostprofile; then timeout -k 10 300 python3 tools/sessions/dist_host_profile_r05.py 2>&1 | grep -v "^fcvm" | tail -40 | cut -c1-200 | tee $O/host_profile.txt; fi
if want tests; then timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_configs.py -q -m gpu -k "boundary or rank or partition" 2>&1 | tail -4; fi
if want rehearsal2; then  # FCVM_DIST_L1_SPLIT / FCVM_DIST_EXCHANGE / FCVM_DIST_K1S are read from the caller's environment
  GPU_MAX_HW_QUEUES=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > $O/bench2.json 2> $O/bench2.err; echo "rc2=$?"; tail -c 600 $O/bench2.json
fi
if want rehearsal6; then
  GPU_MAX_HW_QUEUES=1 FCVM_DIST_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 6 --steps 20 --warmup 5 --no-cpu > $O/bench6.json 2> $O/bench6.err; echo "rc6=$?"; tail -c 600 $O/bench6.json
fi
if want l1parts; then timeout -k 10 200 python3 tools/sessions/l1_pass_parts_r05.py 2>&1 | grep -v "^fcvm" | tail -2; fi
