#!/bin/bash
# round 3: switches of the symmetric form; GNLY probe and Embankment analysis at the final defaults
O=gpurun_out/r03bm
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 400 python3 -m pytest tests -q -m gpu -x -k "set_up_switches" > $O/tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -6 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python3 tools/gnly_probe.py --cells 55 > $O/gnly.txt 2>&1; echo gnly rc=$?; tail -3 $O/gnly.txt
timeout -k 10 300 python3 tools/example_run.py embankment --refine 3 > $O/emb3.txt 2>&1; echo emb rc=$?; tail -1 $O/emb3.txt | cut -c1-330
