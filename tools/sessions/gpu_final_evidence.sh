#!/bin/bash
# evidence of the final build: default bench line, --warmup 20 line, the same command under rocprofv3 with the kernel summary
set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python3 bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err
timeout -k 10 600 python3 bench.py --no-cpu --no-cube --warmup 20 > gpurun_out/final_bench_warmup20.json 2> gpurun_out/final_bench_warmup20.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o a -- python3 $R/bench.py --no-cpu --no-cube > $R/gpurun_out/final_bench_under_rocprof.json 2> $R/gpurun_out/final_bench_under_rocprof.err
f=$(find /tmp/bp -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $R/gpurun_out/final_kernel_stats.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-116s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "average_us", "%"))
for r in rows[:40]:
    print("%-116s %8s %14.1f %12.3f %7s" % (r["Name"][:116], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
P
cd $R
for f in default warmup20 under_rocprof; do python3 -c "
import json,sys; d=json.loads(open('gpurun_out/final_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('achieved'), (d.get('north_star_workload') or {}).get('analysis_s'))"; done
