#!/bin/bash
# round-2 GPU session D: padded-x SpMV A/B + TA counters, bench with kernel trace
O=gpurun_out/r02d
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/spmv_ab.py --cells 55 --rounds 10 --x4 --variants FCVM_SPMV_GANG=1 > $O/ab_x4_cube.log 2>&1 || exit 1
cat $O/ab_x4_cube.log
timeout -k 10 300 python3 tools/spmv_ab.py --mesh embankment3 --rounds 10 --x4 --variants FCVM_SPMV_GANG=1 > $O/ab_x4_emb.log 2>&1 || exit 1
cat $O/ab_x4_emb.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "pcg_solve or both_precond or cube_analysis or embankment_analysis or bitwise or u100_column_full" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -3 $O/parity.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_x4.log 2>&1 || exit 1
FCVM_SPMV_XPAD=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_x3.log 2>&1 || exit 1
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*' $O/bench_x4.log $O/bench_x3.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_trace.log 2>&1 || exit 1
timeout -k 10 240 rocprofv3 --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_ta -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-plate > $O/pmc_ta.log 2>&1
echo done
