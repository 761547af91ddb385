#!/bin/bash
# round 4: is the host-sequenced distributed cycle host-bound or GPU-bound?  kernel time per PCG iteration under rocprofv3
set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp
FORCE_MODES=1 timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -o a -- python3 $R/tools/sessions/dist_host_overhead_r04.py > $R/gpurun_out/r04za_run.log 2>&1
f=$(find /tmp/dp -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $R/gpurun_out/r04za_kernel_stats.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:40]:
    print("%-90s %8s %12.1f %10.3f %6s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
P
cd $R
grep "host-sequenced" gpurun_out/r04za_run.log
head -45 gpurun_out/r04za_kernel_stats.txt | cut -c1-70,90-140
