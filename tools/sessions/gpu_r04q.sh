#!/bin/bash
# round 4 evidence: default bench, bench under rocprofv3 --kernel-trace --stats, PMC traffic of the FP64 SpMV (symmetric form) and of
# the assembly into the symmetric form
set -o pipefail
O=gpurun_out/r04q; mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
python bench.py --warmup 20 --no-cpu --no-cube > $O/bench_warmup20.json 2> $O/bench_warmup20.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o b -- python3 $R/bench.py --no-cpu --no-cube > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_kernel_stats.csv; rm -rf $O/prof
bash tools/pmc_lo.sh $O/pmc_spmv > $O/pmc_spmv.txt 2>&1; tail -5 $O/pmc_spmv.txt
bash tools/pmc_asm.sh $O/pmc_asm > $O/pmc_asm.txt 2>&1; tail -5 $O/pmc_asm.txt
rm -rf $O/pmc_spmv/*/ $O/pmc_asm/*/ 2>/dev/null
