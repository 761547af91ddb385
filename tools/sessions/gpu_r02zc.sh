#!/bin/bash
O=gpurun_out/r02zc
mkdir -p $O
export TMPDIR=/tmp
for d in 6 10 15; do
FCVM_RHS_PROJECTION=$d timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench_p$d.log 2>&1 || { tail $O/bench_p$d.log; exit 1; }
echo "projection depth $d"; grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*' $O/bench_p$d.log
done
for d in 6 12; do
FCVM_RHS_PROJECTION=$d timeout -k 10 300 python3 bench.py --steps 40 --warmup 20 --no-cpu --no-plate > $O/bench_long_p$d.log 2>&1
echo "projection depth $d, steps 40 after 20"; grep -o '"ms_per_step": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*' $O/bench_long_p$d.log
done
