#!/bin/bash
# round-2 GPU session C: LDS assembly kernel parity + timing, Lug pressure pin, PMC passes (new and old assembly kernel)
set -o pipefail
O=gpurun_out/r02c
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "assemble or tangent or lug_pressure or cube_analysis or embankment_analysis or gnly_updated or buckling_factors or size_independent or bitwise" > $O/parity.log 2>&1
rc=$?; tail -n 15 $O/parity.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-plate > $O/bench_lds.log 2>&1 || exit 1
FCVM_ASSEMBLE_LDS=0 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-plate > $O/bench_rows.log 2>&1 || exit 1
grep -o '"assembly": {[^}]*}' $O/bench_lds.log $O/bench_rows.log
bash tools/pmc_probe.sh $O/pmc_lds || exit 1
export FCVM_ASSEMBLE_LDS=0
bash tools/pmc_probe.sh $O/pmc_rows
