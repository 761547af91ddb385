#!/bin/bash
# round 5: what the REPLICATED sparse aggregate level costs per PCG iteration at the 6-rank size: ONE rank holding the whole 6.0 M mesh through the
# host-sequenced layer (FCVM_DIST_FORCE_HOST=1 -> DistMG with 13 k aggregates, sparse level + super-aggregates), per-iteration kernel table
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dq
PLATE_H=0.79796 FORCE_MODES=1 FCVM_VERBOSE=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/dq -- python3 $R/tools/sessions/dist_host_overhead_r04.py > $O/run.log 2>&1; echo "rc=$?"
cd $R
grep -E "host-sequenced|fcvm dist: aggregate|sparse aggregate" $O/run.log | cut -c1-300
python3 tools/sessions/steady_state_stats_r05.py /tmp/dq k_dist_cg_update 60 150 | tee $O/dist_cycle_6M.txt | cut -c1-150
