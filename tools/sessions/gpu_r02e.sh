#!/bin/bash
# round-2 GPU session E: assembly v2 + split level-1 SpMV + fused restrict: parity, kernel trace of the bench, A/B
O=gpurun_out/r02e
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "dense_spd or assemble or tangent or lug_pressure or cube_analysis or embankment_analysis or gnly_updated or buckling_factors or bitwise or pcg_solve or both_precond or refined_embankment" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -3 $O/parity.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_trace.log 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_new.log 2>&1 || exit 1
FCVM_SPMV_SPLIT1=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_nosplit.log 2>&1 || exit 1
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"assembly": {[^}]*}' $O/bench_new.log $O/bench_nosplit.log
head -30 $O/trace/runc/*kernel_stats.csv | cut -c1-60,200-400
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py -x -q > $O/dist.log 2>&1; tail -3 $O/dist.log
