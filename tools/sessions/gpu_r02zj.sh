#!/bin/bash
# the largest single-GPU configuration claimed in DESIGN.md: 12 M tet10 (bench.py --cells 126)
O=gpurun_out/r02zj
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
FCVM_VERBOSE=1 timeout -k 10 700 python3 bench.py --cells 126 --steps 3 --warmup 1 --no-cpu --no-plate > $O/bench12m.log 2> $O/bench12m.err; echo "rc $?"
grep "window format\|sparse aggregate" $O/bench12m.err | head -3
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"setup_s": {[^}]*}\|aggregates x[^"]*\|"workload": "[^"]*"' $O/bench12m.log
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
