#!/bin/bash
# round 5: the multi-GPU layer's collectives over RCCL (backend nccl) on ONE rank -- the only RCCL run this pool allows (two ranks on one GPU are refused by RCCL):
# process group, all-reduces of the PCG on device tensors, object collectives of the set-up
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
for b in gloo nccl; do
  echo "== backend $b" | tee -a $O/rccl_one_rank.txt
  FCVM_DIST_BACKEND=$b FORCE_MODES=1 timeout -k 10 300 python3 tools/sessions/dist_host_overhead_r04.py 2>&1 | grep -E "host-sequenced|Error|error|NCCL|RCCL" | tail -4 | cut -c1-300 | tee -a $O/rccl_one_rank.txt
done
