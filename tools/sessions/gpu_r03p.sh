#!/bin/bash
O=gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python3 tools/pair_debug.py emb > $O/pair_emb.log 2>&1; echo rc=$?; cat $O/pair_emb.log | tail -14
timeout -k 10 200 python3 tools/pair_debug.py cube > $O/pair_cube.log 2>&1; echo rc=$?; cat $O/pair_cube.log | tail -14
FCVM_VERBOSE=1 timeout -k 10 300 python3 tools/gnly_probe.py --steps 3 > $O/gnly_pair.log 2>&1; echo pair rc=$?; grep -i "solve2\|iteration\|window" $O/gnly_pair.log | tail -8
