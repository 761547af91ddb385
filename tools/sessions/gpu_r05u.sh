#!/bin/bash
# round 5: the dense super-aggregate level of the replicated sparse aggregate level at the 6-rank size (ONE rank holding the 6.0 M mesh, 13 k aggregates): how
# many super-aggregates (FCVM_L2S_MAX_SUPER) against PCG iterations and time per iteration -- its FP32 inverse is streamed twice per cycle by EVERY rank
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05u; mkdir -p $O; rm -f $O/scan.txt
cd $R
for ms in 2000 900 450 230; do
  echo "== FCVM_L2S_MAX_SUPER=$ms (FCVM_MG_TARGET=69, 2 steps per sweep, 2 corrections)" | tee -a $O/scan.txt
  PLATE_H=0.79796 FORCE_MODES=1 FCVM_VERBOSE=1 FCVM_MG_TARGET=69 FCVM_L2S_MAX_SUPER=$ms timeout -k 10 300 python3 tools/sessions/dist_host_overhead_r04.py 2>&1 | grep -E "host-sequenced|fcvm dist: aggregate" | tail -2 | cut -c1-300 | tee -a $O/scan.txt
done
