#!/bin/bash
# round 5: RCCL on a one-GPU box: (a) the multi-GPU layer's collectives on ONE rank (backend nccl), (b) point-to-point as the Exchanger issues it, the rank itself as the peer
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z3; mkdir -p $O
cd $R
timeout -k 10 120 python3 tools/sessions/rccl_p2p_self_r05.py 2>&1 | grep -v "^\[W\|Warning" | tail -6 | cut -c1-300 | tee $O/rccl_p2p_self.txt
