#!/bin/bash
# round 5: the multi-GPU cycle on the current kernels -- GPU tests of the distributed layer and of the boundary-first split, then ONE rank on the 1 M
# plate through the host-sequenced layer (FCVM_DIST_FORCE_HOST=1) against the library's own solver, and its kernel statistics under rocprofv3
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -x -q -m gpu -k "ranks or partition or buckling or boundary_groups or group_map" > $O/tests.log 2>&1; rc=$?
tail -3 $O/tests.log | cut -c1-300
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $O/tests.log | head -30 | cut -c1-300; exit 1; }
timeout -k 10 300 python3 tools/sessions/dist_host_overhead_r04.py 2>&1 | grep "host-sequenced" | tee $O/host_overhead.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp
FORCE_MODES=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -o a -- python3 $R/tools/sessions/dist_host_overhead_r04.py > $O/run.log 2>&1
f=$(find /tmp/dp -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $O/kernel_stats.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:48]:
    print("%-90s %8s %12.1f %10.3f %6s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
P
cd $R
grep "host-sequenced" $O/run.log
head -50 $O/kernel_stats.txt | cut -c1-70,90-140
