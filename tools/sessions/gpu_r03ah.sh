#!/bin/bash
# round 3: element map built on the device, Morton sort on all threads -- parity test, set-up timing, default bench
O=gpurun_out/r03ah
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python3 -m pytest tests -q -m gpu -k "element_map or assemble or tangent or window_format or fcVM_mirror" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
for mt in 1 4 8 16; do
  FCVM_MORTON_THREADS=$mt FCVM_PATTERN_TIMING=1 timeout -k 10 200 python3 tools/setup_inside.py > $O/setup_mt$mt.txt 2>&1 || exit 1
  echo "== Morton threads $mt"; grep "set_mesh:" $O/setup_mt$mt.txt | head -12
done
FCVM_EMAP_HOST=1 FCVM_PATTERN_TIMING=1 timeout -k 10 200 python3 tools/setup_inside.py > $O/setup_emaphost.txt 2>&1; echo "== host emap"; grep "set_mesh:" $O/setup_emaphost.txt | head -12
timeout -k 10 400 python3 bench.py --no-cpu --no-cube > $O/bench.json 2> $O/bench.err; echo bench rc=$?
python3 -c "
import json
d=json.loads(open('$O/bench.json').read()); print(round(d['ms_per_step'],2), d['pcg_iterations_per_solve'], round(d['roofline']['frac'],3), d['setup_s'], round(d['north_star_workload']['analysis_s'],3))"
