#!/bin/bash
# round 3: symmetric window form -- parity subset after the builder / copy changes, set-up times, HBM traffic of the kernel (PMC)
O=gpurun_out/r03bh
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 400 python3 -m pytest tests -q -m gpu -x -k "symmetric_window or window_format_spmv or lockstep or reruns or buckling_factors or set_up_switches" > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -5 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
FCVM_PATTERN_TIMING=1 timeout -k 10 200 python3 tools/sessions/gm_prof_r03.py > $O/gm.txt 2>&1; grep "GpuModel\|set_mesh:" $O/gm.txt | tail -8
PROBE="python3 tools/lo_probe.py --reps 4 --solves 1"
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $cnt --output-format csv -d $O/pmc_$cnt -- $PROBE > $O/pmc_$cnt.log 2>&1 || { echo "pmc $cnt failed"; exit 1; }
done
python3 tools/pmc_summary.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete
grep "stream_read\|spmv_win" $O/summary.txt | cut -c1-150
