#!/bin/bash
# round-2 GPU session Q: window-format SpMV with split rows; why are 3 ranks on one GPU slow
O=gpurun_out/r02q
mkdir -p $O
export TMPDIR=/tmp
FCVM_VERBOSE=1 timeout -k 10 300 python3 tools/spmv_ab.py --cells 55 --rounds 10 --variants FCVM_SPMV_WIN=0 FCVM_SPMV_WIN=1 > $O/ab_win.log 2>&1 || { tail -20 $O/ab_win.log; exit 1; }
grep -v "candidate" $O/ab_win.log | cut -c1-250
FCVM_VERBOSE=1 timeout -k 10 300 python3 tools/spmv_ab.py --mesh embankment3 --rounds 10 --variants FCVM_SPMV_WIN=0 FCVM_SPMV_WIN=1 > $O/ab_win_emb.log 2>&1 || { tail -20 $O/ab_win_emb.log; exit 1; }
grep -v "candidate" $O/ab_win_emb.log | cut -c1-250
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "pcg or precond or spmv or cube or lug or embankment" > $O/parity.log 2>&1; tail -3 $O/parity.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench.log 2>&1 || { tail $O/bench.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"back_to_back_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"stream_read_GBs_this_box": [0-9.]*\|"ms": [0-9.]*' $O/bench.log
timeout -k 10 150 python3 tools/dist_debug.py --ranks 3 --refine 1 --nstep 1 --tag r02q_base > $O/dd_base.log 2>&1; echo "base rc $?"; tail -1 $O/../r02q_base/dist_rank0.log | cut -c1-200
OMP_NUM_THREADS=1 timeout -k 10 150 python3 tools/dist_debug.py --ranks 3 --refine 1 --nstep 1 --tag r02q_omp1 > $O/dd_omp1.log 2>&1; echo "omp1 rc $?"; tail -1 $O/../r02q_omp1/dist_rank0.log | cut -c1-200
GPU_MAX_HW_QUEUES=1 timeout -k 10 150 python3 tools/dist_debug.py --ranks 3 --refine 1 --nstep 1 --tag r02q_hwq1 > $O/dd_hwq1.log 2>&1; echo "hwq1 rc $?"; tail -1 $O/../r02q_hwq1/dist_rank0.log | cut -c1-200
timeout -k 10 150 python3 tools/dist_debug.py --ranks 2 --refine 1 --nstep 1 --tag r02q_two > $O/dd_two.log 2>&1; echo "two rc $?"; tail -1 $O/../r02q_two/dist_rank0.log | cut -c1-200
