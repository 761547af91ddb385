#!/bin/bash
# round-2 GPU session R: window-format SpMV variants (waves per group, loads before staging), row order and the repack,
# multi-rank tests with one hardware queue per rank
O=gpurun_out/r02r
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python3 tools/spmv_ab.py --cells 55 --rounds 8 --variants FCVM_SPMV_WIN=0 FCVM_SPMV_WIN=1 FCVM_WIN_PRE=1 FCVM_WIN_WAVES=4 FCVM_WIN_WAVES=4,FCVM_WIN_PRE=1 > $O/ab_win.log 2>&1 || { tail -20 $O/ab_win.log; exit 1; }
cat $O/ab_win.log | cut -c1-250
timeout -k 10 400 python3 tools/spmv_ab.py --mesh embankment3 --rounds 8 --variants FCVM_SPMV_WIN=0 FCVM_SPMV_WIN=1 FCVM_WIN_PRE=1 FCVM_WIN_WAVES=4 FCVM_WIN_WAVES=4,FCVM_WIN_PRE=1 > $O/ab_win_emb.log 2>&1 || { tail -20 $O/ab_win_emb.log; exit 1; }
cat $O/ab_win_emb.log | cut -c1-250
for ro in 0 1; do
FCVM_ROW_ORDER=$ro timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_ro$ro.log 2>&1 || { tail $O/bench_ro$ro.log; exit 1; }
echo "row order $ro"; grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"assembly": {"ms": [0-9.]*' $O/bench_ro$ro.log
done
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
PYTHONUNBUFFERED=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py tests/test_gpu_configs.py -x -v --durations=0 -k "not config4" > $O/dist.log 2>&1; echo "dist rc $?"
grep -E "passed|failed" $O/dist.log | tail -2; grep -E "^[0-9.]+s call" $O/dist.log | head -12
