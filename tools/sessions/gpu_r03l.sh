#!/bin/bash
# round 3, call l: default bench.py under rocprofv3 --kernel-trace --stats (profiles/), traffic counters of the plate SpMV
O=gpurun_out/r03l
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo bench rc=$?
timeout -k 10 300 python3 bench.py --warmup 20 --steps 20 --no-cpu --no-cube > $O/bench_w20.json 2> $O/bench_w20.err; echo bench w20 rc=$?
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python3 $R/bench.py --no-cpu --no-cube > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err; echo rocprof rc=$?
cd $R
python3 tools/rocpd_top.py $O/prof/bench_results.db 40 > $O/bench_kernel_stats.txt
rm -f $O/prof/*.db
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/lo_probe.py --reps 2 --solves 1 > $O/pmc_$c.log 2>&1; echo pmc $c rc=$?
done
python3 - <<'PY' > gpurun_out/r03l/traffic.txt
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r03l/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if "spmv_win" in k or "stream_read" in k:
            print(c, k, "n=%d mean=%.1f min=%.1f max=%.1f" % (len(v), sum(v) / len(v), min(v), max(v)))
PY
cat $O/traffic.txt; grep "plain SpMV\|stream" $O/pmc_FETCH_SIZE.log
find $O -name "*.csv" -size +1M -delete
