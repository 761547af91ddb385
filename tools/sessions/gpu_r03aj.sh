#!/bin/bash
# round 3: level-1 topology on a background thread since set_mesh, level-2 host loops on all threads, device start vector of the
# power iterations -- stage times, parity subset, bench
O=gpurun_out/r03aj
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
FCVM_PATTERN_TIMING=1 FCVM_SETUP_TIMING=1 timeout -k 10 200 python3 tools/setup_inside.py > $O/setup.txt 2>&1; grep "fcvm set-up\|RiksDriver\|set_mesh:" $O/setup.txt
FCVM_TOPO_EARLY=0 FCVM_SETUP_TIMING=1 timeout -k 10 200 python3 tools/setup_inside.py > $O/setup_noearly.txt 2>&1; grep "fcvm set-up:   level-1 topo\|RiksDriver" $O/setup_noearly.txt
timeout -k 10 600 python3 -m pytest tests -q -m gpu -x -k "preconditioners or 16bit or lockstep or window_format or sparse_aggregate or reruns or size_independent or embankment or fcVM_mirror or element_map" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
timeout -k 10 400 python3 bench.py --no-cpu --no-cube > $O/bench.json 2> $O/bench.err; echo bench rc=$?
python3 -c "
import json
d=json.loads(open('$O/bench.json').read()); print(round(d['ms_per_step'],2), d['pcg_iterations_per_solve'], round(d['roofline']['frac'],3), d['setup_s'], round(d['north_star_workload']['analysis_s'],3), d['north_star_workload']['setup_s_inside'])"
