#!/bin/bash
# round-2 GPU session L: why did the 3-rank embankment test stall; SpMV inside the solver vs back to back
O=gpurun_out/r02l
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/dist_debug.py --ranks 3 --refine 1 --nstep 8 --tag r02l > $O/dist_debug.log 2>&1; echo "dist_debug rc $?"; tail -3 $O/dist_debug.log
for r in 0 1 2; do grep -c "^solve" $O/dist_rank$r.log; tail -4 $O/dist_rank$r.log | cut -c1-300; done
FCVM_VERBOSE=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $O/bench.log 2>&1 || { tail $O/bench.log; exit 1; }
grep "candidate" $O/bench.log
grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*\|"back_to_back_launch_ms": [0-9.]*\|"frac": [0-9.]*\|"pcg_iterations_per_step": [0-9.]*\|"stream_read_GBs_this_box": [0-9.]*' $O/bench.log
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-plate > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT && find $O/prof -name "*kernel_stats.csv" | while read f; do head -12 "$f" | cut -c1-200; done
echo done
