#!/bin/bash
# round 3: HBM traffic of the three symmetric fine-level passes (PMC FETCH_SIZE / WRITE_SIZE, separate passes)
O=gpurun_out/r03br
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
PROBE="python3 tools/lo_probe.py --reps 4 --solves 1"
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $cnt --output-format csv -d $O/pmc_$cnt -- $PROBE > $O/pmc_$cnt.log 2>&1 || { echo "pmc $cnt failed"; exit 1; }
done
find $O -name "*.csv" -size +2M -delete
ls $O/pmc_*/*/ | head
