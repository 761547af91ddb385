#!/bin/bash
# scan of two cycle parameters on the bench workload (weight of the Jacobi part of the additive cycle, level-1 Chebyshev interval)
O=gpurun_out/r02zi
mkdir -p $O
export TMPDIR=/tmp
for w in 0.7 1.0 1.4; do for al in 4 8 16; do
FCVM_MG_ADD_W=$w FCVM_MG_ALPHA1S=$al timeout -k 10 200 python3 bench.py --no-cpu --no-plate > $O/b_${w}_${al}.log 2>&1 || { tail -3 $O/b_${w}_${al}.log; continue; }
echo "w $w alpha1s $al: $(grep -o '"ms_per_step": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*' $O/b_${w}_${al}.log | tr '\n' ' ')"
done; done
