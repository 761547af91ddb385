#!/bin/bash
# round 5, final build: default bench line and the same command under rocprofv3 --kernel-trace --stats (as gpu_r05p.sh, without the counter passes)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05zz; mkdir -p $O
cd $R
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o a -- python3 $R/bench.py --no-cpu --no-cube > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
f=$(find /tmp/bp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/bench_kernel_stats.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-116s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "average_us", "%"))
for r in rows[:44]:
    print("%-116s %8s %14.1f %12.3f %7s" % (r["Name"][:116], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
P
cd $R
for f in default under_rocprof; do python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r05zz/bench_$f.json').read().strip().splitlines()[-1]); print('$f', 'value %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], 'avg_launch_ms %.4f' % d['roofline']['avg_launch_ms'], 'analysis_s', (d.get('north_star_workload') or {}).get('analysis_s'), 'tuning', d['setup_s'].get('placement_tuning'), 'gnly', (d.get('gnly') or {}).get('ms_per_newton_iteration'), 'asm', d['assembly']['ms'])"; done
head -16 $O/bench_kernel_stats.txt | cut -c1-60,117-170
