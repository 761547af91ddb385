#!/bin/bash
# round-2 GPU session G: shipped SpMV (8+16 byte gathers) vs three 8-byte gathers with TA counters, traffic of the shipped
# kernel, assembly classes side by side vs one after the other, kernel trace of the bench
O=gpurun_out/r02g
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/spmv_ab.py --cells 55 --rounds 10 --variants FCVM_SPMV_VARIANT=5 FCVM_SPMV_VARIANT=4 > $O/ab_gather.log 2>&1 || exit 1
cat $O/ab_gather.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "assemble or tangent or pcg_solve or both_precond or cube_analysis or embankment_analysis or bitwise or mirror_api or gnly_updated or buckling_factors" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
for v in 5 4; do
  export FCVM_SPMV_VARIANT=$v
  timeout -k 10 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_ta_v$v -- python3 tools/spmv_probe.py --reps 10 --pcg 0 > $O/pmc_ta_v$v.log 2>&1
done
export FCVM_SPMV_VARIANT=5
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/spmv_probe.py --reps 10 --pcg 0 > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/spmv_probe.py --reps 10 --pcg 0 > $O/pmc_write.log 2>&1
unset FCVM_SPMV_VARIANT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-plate > $O/bench_trace.log 2>&1 || exit 1
FCVM_ASSEMBLE_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_seq -- python3 tools/spmv_probe.py --reps 2 --pcg 0 > $O/trace_seq.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_ovl -- python3 tools/spmv_probe.py --reps 2 --pcg 0 > $O/trace_ovl.log 2>&1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench.log 2>&1 || exit 1
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"stream_read_GBs_this_box": [0-9.]*' $O/bench.log
echo done
