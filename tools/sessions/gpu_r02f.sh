#!/bin/bash
# round-2 GPU session F: whole GPU test suite, default bench line, SpMV traffic counters
O=gpurun_out/r02f
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_gpu_configs.py -k "not elastica" > $O/pytest_gpu.log 2>&1
rc=$?; tail -25 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python3 bench.py > $O/bench_default.log 2>&1 || { tail -5 $O/bench_default.log; exit 1; }
tail -c 3000 $O/bench_default.log
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/spmv_probe.py --reps 10 --pcg 0 > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/spmv_probe.py --reps 10 --pcg 0 > $O/pmc_write.log 2>&1
echo done
timeout -k 10 300 python3 tools/spmv_ab.py --cells 55 --rounds 10 --variants FCVM_SPMV_VARIANT=4 FCVM_SPMV_VARIANT=5 > $O/ab_gather.log 2>&1; cat $O/ab_gather.log
timeout -k 10 300 python3 tools/spmv_ab.py --mesh embankment3 --rounds 10 --variants FCVM_SPMV_VARIANT=4 FCVM_SPMV_VARIANT=5 > $O/ab_gather_emb.log 2>&1; cat $O/ab_gather_emb.log
