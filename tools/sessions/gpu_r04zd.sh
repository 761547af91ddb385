#!/bin/bash
# round 4 experiment: window groups of 16 waves (1024 rows) instead of 8 -- stored fraction of the symmetric form, SpMV and bench time
# the A/B library is not tracked (*.so): build it first, here or on the GPU box (same image, hipcc present):
#   make -C fcvm-workbench_amd/csrc OUT=../../tools/sessions/ab/libfcvm_w16.so EXTRA=-DFCVM_WIN_WAVES=16
mkdir -p gpurun_out
export FCVM_HIP_LIB=$GRAFT_REPO_ROOT/tools/sessions/ab/libfcvm_w16.so
if [ ! -f "$FCVM_HIP_LIB" ]; then
  mkdir -p $GRAFT_REPO_ROOT/tools/sessions/ab
  make -C $GRAFT_REPO_ROOT/fcvm-workbench_amd/csrc OUT=../../tools/sessions/ab/libfcvm_w16.so EXTRA=-DFCVM_WIN_WAVES=16 > /dev/null || { echo "could not build $FCVM_HIP_LIB"; exit 1; }
fi
[ -f "$FCVM_HIP_LIB" ] || { echo "missing $FCVM_HIP_LIB"; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "symmetric_window_form_spmv or window_format_spmv or straight_into" > gpurun_out/r04zd_tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r04zd_tests.log | cut -c1-200
timeout -k 10 600 python3 bench.py --no-cpu --no-cube > gpurun_out/r04zd_bench.json 2> gpurun_out/r04zd_bench.err; echo "bench rc $?"
tail -3 gpurun_out/r04zd_bench.err | cut -c1-300
python3 -c "
import json; d=json.loads(open('gpurun_out/r04zd_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], d['pcg_iterations_per_step'], r['storage'], r['stored_blocks'], r['matrix_blocks'], r['avg_launch_ms'], r['frac'], d['assembly'])"
unset FCVM_HIP_LIB
timeout -k 10 600 python3 bench.py --no-cpu --no-cube > gpurun_out/r04zd_bench8.json 2> gpurun_out/r04zd_bench8.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r04zd_bench8.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], d['pcg_iterations_per_step'], r['storage'], r['stored_blocks'], r['matrix_blocks'], r['avg_launch_ms'], r['frac'], d['assembly'])"
