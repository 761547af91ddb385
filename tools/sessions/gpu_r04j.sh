#!/bin/bash
# GNLY Newton iteration on the 1 M cube: stage times of the tangent refresh, kernel stats
set -o pipefail
O=gpurun_out/r04j; mkdir -p $O
FCVM_SETUP_TIMING=1 python tools/gnly_probe.py --cells 55 --steps 4 > $O/gnly_stages.log 2>&1; tail -40 $O/gnly_stages.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o g -- python3 $R/tools/gnly_probe.py --cells 55 --steps 6 > $R/$O/gnly_prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/gnly_kernel_stats.csv; rm -rf $O/prof
grep iteration $O/gnly_prof.log
python3 - <<'P'
import csv
rows = list(csv.DictReader(open("gpurun_out/r04j/gnly_kernel_stats.csv")))
for r in rows[:28]:
    print("%-70s calls %5s avg %9.1f us total %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
P
