#!/bin/bash
# round-2 GPU session Y: multiplicative V(1,1) cycle against the additive one, now that the fine SpMV is cheaper
O=gpurun_out/r02y
mkdir -p $O
export TMPDIR=/tmp
for add in 1 0; do
FCVM_MG_ADDITIVE=$add timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench_add$add.log 2>&1 || { tail $O/bench_add$add.log; exit 1; }
echo "additive $add"; grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"fine_spmv_per_step": [0-9.]*' $O/bench_add$add.log
done
