#!/bin/bash
# round 4: kernel stats of the bench (level-1 SpMV with eight records in flight) + the bench line
set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o a -- python3 $R/bench.py --no-cpu --no-cube > $R/gpurun_out/r04z_bench_prof.json 2> $R/gpurun_out/r04z_bench_prof.err
f=$(find /tmp/bp -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $R/gpurun_out/r04z_kernel_stats.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-110s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "average_us", "%"))
for r in rows[:26]:
    print("%-110s %8s %12.1f %10.3f %6s" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
P
cd $R
head -12 gpurun_out/r04z_kernel_stats.txt | cut -c1-60,100-160
timeout -k 10 600 python3 bench.py --no-cpu --no-cube > gpurun_out/r04z_bench.json 2> gpurun_out/r04z_bench.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r04z_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('pcg_iterations_per_step'))"
