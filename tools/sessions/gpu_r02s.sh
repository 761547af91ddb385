#!/bin/bash
# round-2 GPU session S: window format as default (Morton row order for the assembly layout): parity, multi-rank tests,
# kernel stats (assembly + repack), counters of the window kernel, clock log
O=gpurun_out/r02s
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "not elastica and not plate_with_hole and not bench_line" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
PYTHONUNBUFFERED=1 timeout -k 10 500 python3 -m pytest tests/test_gpu_dist.py -x -v --durations=0 > $O/dist.log 2>&1; echo "dist rc $?"
grep -E "passed|failed" $O/dist.log | tail -2; grep -E "^[0-9.]+s call" $O/dist.log | head -8
for ro in 1 0; do
FCVM_ROW_ORDER=$ro timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ro$ro -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_ro$ro.log 2>&1 || { tail $O/bench_ro$ro.log; exit 1; }
echo "row order $ro"; grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"assembly": {"ms": [0-9.]*' $O/bench_ro$ro.log
f=$(find $O/trace_ro$ro -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-160
done
PROBE="python3 tools/spmv_probe.py --reps 6 --pcg 0 --membench"
run() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc/$name -- $PROBE > $O/pmc_$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name hung: stopping"; exit 1; fi
}
mkdir -p $O/pmc
timeout -k 10 240 python3 tools/spmv_probe.py --reps 30 --pcg 0 --membench > $O/pmc/plain.log 2>&1 || exit 1
run fetch FETCH_SIZE
run write WRITE_SIZE
run ta1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
run grbm GRBM_GUI_ACTIVE
run tcc1 TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
cat $O/pmc_summary.txt | cut -c1-160
timeout -k 10 200 python3 tools/clock_log.py --what spmv --seconds 12 > $O/clock_spmv.log 2>&1; head -3 $O/clock_spmv.log | cut -c1-300; tail -25 $O/clock_spmv.log
