#!/bin/bash
# round-2 GPU session K: matrix storage without pair rounding -- parity, SpMV timing, bench
O=gpurun_out/r02k
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "not elastica and not plate_with_hole and not bench_line" > $O/parity.log 2>&1 || { tail -40 $O/parity.log; exit 1; }
tail -2 $O/parity.log
timeout -k 10 300 python3 tools/spmv_ab.py --cells 55 --rounds 10 --variants FCVM_SPMV_VARIANT=5 FCVM_SPMV_VARIANT=4 > $O/ab.log 2>&1 || { tail $O/ab.log; exit 1; }
cat $O/ab.log
timeout -k 10 300 python3 tools/spmv_ab.py --mesh embankment3 --rounds 10 --variants FCVM_SPMV_VARIANT=5 > $O/ab_emb.log 2>&1; cat $O/ab_emb.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench.log 2>&1 || exit 1
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"stream_read_GBs_this_box": [0-9.]*\|"padded_matrix_bytes": [0-9]*' $O/bench.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py -x -q > $O/dist.log 2>&1; tail -2 $O/dist.log
echo done
