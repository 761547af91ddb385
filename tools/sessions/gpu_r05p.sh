#!/bin/bash
# round 5 evidence of the final build: default bench line, --warmup 20 line, the same command under rocprofv3 --kernel-trace --stats, PMC traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes, no trace domains) of the FP64 SpMV and the 16-bit passes with the chunked group map
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd $R
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 400 python3 bench.py --no-cpu --no-cube --warmup 20 > $O/bench_warmup20.json 2> $O/bench_warmup20.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -o a -- python3 $R/bench.py --no-cpu --no-cube > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
f=$(find /tmp/bp -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $O/bench_kernel_stats.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-116s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "average_us", "%"))
for r in rows[:44]:
    print("%-116s %8s %14.1f %12.3f %7s" % (r["Name"][:116], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
P
cd $R
for f in default warmup20 under_rocprof; do python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r05p/bench_$f.json').read().strip().splitlines()[-1]); print('$f', 'value %.4g' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], 'avg_launch_ms %.4f' % d['roofline']['avg_launch_ms'], 'analysis_s', (d.get('north_star_workload') or {}).get('analysis_s'), 'tuning', d['setup_s'].get('placement_tuning'), 'gnly', (d.get('gnly') or {}).get('ms_per_newton_iteration'), 'asm', d['assembly']['ms'])"; done
head -16 $O/bench_kernel_stats.txt | cut -c1-60,117-170
mkdir -p $O/pmc
PROBE="python3 tools/lo_probe.py --reps 4 --solves 1"
timeout -k 10 240 $PROBE > $O/pmc/plain.log 2>&1
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $cnt --output-format csv -d $O/pmc/$cnt -- $PROBE > $O/pmc/$cnt.log 2>&1; echo "$cnt rc=$?"
done
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
find $O/pmc -name "*.csv" -size +1M -delete
grep -A1 -E "k_spmv_win_sym<true|k_spmv_win_lo_sym<[12]|k_stream_read " $O/pmc_summary.txt | head -40
