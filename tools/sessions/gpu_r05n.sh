#!/bin/bash
# round 5: per-iteration kernel table of the host-sequenced multi-GPU cycle (ONE rank, FCVM_DIST_FORCE_HOST=1, 1 M plate) and of the library's own loop
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp1 /tmp/dp0
FORCE_MODES=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/dp1 -- python3 $R/tools/sessions/dist_host_overhead_r04.py > $O/run1.log 2>&1
FORCE_MODES=0 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/dp0 -- python3 $R/tools/sessions/dist_host_overhead_r04.py > $O/run0.log 2>&1
cd $R
grep "host-sequenced" $O/run1.log $O/run0.log
python3 tools/sessions/steady_state_stats_r05.py /tmp/dp1 k_dist_cg_update 55 140 | tee $O/dist_cycle.txt | cut -c1-150
python3 tools/sessions/steady_state_stats_r05.py /tmp/dp0 k_pcg_update_p 55 130 | tee $O/lib_cycle.txt | cut -c1-150
