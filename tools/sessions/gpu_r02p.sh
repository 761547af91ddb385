#!/bin/bash
# round-2 GPU session P: window-format SpMV against the sliced-ELL kernel (same process), parity subset, 3-rank profile
O=gpurun_out/r02p
mkdir -p $O
export TMPDIR=/tmp
FCVM_VERBOSE=1 timeout -k 10 300 python3 tools/spmv_ab.py --cells 55 --rounds 10 --variants FCVM_SPMV_WIN=0 FCVM_SPMV_WIN=1 > $O/ab_win.log 2>&1 || { tail -20 $O/ab_win.log; exit 1; }
grep -v "candidate" $O/ab_win.log | cut -c1-250
FCVM_VERBOSE=1 timeout -k 10 300 python3 tools/spmv_ab.py --mesh embankment3 --rounds 10 --variants FCVM_SPMV_WIN=0 FCVM_SPMV_WIN=1 > $O/ab_win_emb.log 2>&1 || { tail -20 $O/ab_win_emb.log; exit 1; }
grep -v "candidate" $O/ab_win_emb.log | cut -c1-250
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "pcg or precond or spmv or cube or lug or embankment" > $O/parity.log 2>&1; tail -3 $O/parity.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench.log 2>&1 || { tail $O/bench.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"back_to_back_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"stream_read_GBs_this_box": [0-9.]*\|"ms": [0-9.]*' $O/bench.log
timeout -k 10 200 python3 tools/dist_debug.py --ranks 3 --refine 1 --nstep 2 --tag r02p --profile > $O/dist_debug.log 2>&1; echo "dist_debug rc $?"
grep -A36 "Ordered by" $O/dist_rank0.log | cut -c1-160
