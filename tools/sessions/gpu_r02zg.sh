#!/bin/bash
O=gpurun_out/r02zg
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "not elastica and not bench_line" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
for i in 1 2; do
timeout -k 10 300 python3 bench.py --no-cpu --no-plate > $O/bench$i.log 2>&1 || { tail $O/bench$i.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"fine_spmv_per_step": [0-9.]*' $O/bench$i.log
done
timeout -k 10 300 python3 tools/example_run.py plate --h 1.45 > $O/plate.log 2>&1; tail -4 $O/plate.log | cut -c1-300
