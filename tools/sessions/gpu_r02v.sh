#!/bin/bash
# round-2 GPU session V: level-1 matrix of the multigrid cycle on the window format (FP32 values, Chebyshev epilogue)
O=gpurun_out/r02v
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "window or pcg or precond or cube or lug or embankment or properties" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
for w1 in 1 0 1 0; do
FCVM_SPMV_WIN1=$w1 FCVM_VERBOSE=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench_w$w1.log 2>&1 || { tail $O/bench_w$w1.log; exit 1; }
echo "level-1 window $w1"; grep "level-1 window" $O/bench_w$w1.log | head -1; grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*' $O/bench_w$w1.log
done
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $GRAFT_REPO_ROOT/$O/bench_trace.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -22 "$f" | cut -c1-150
