#!/bin/bash
# round 5, last build: HBM traffic by counters (FETCH_SIZE / WRITE_SIZE in separate passes, no trace domains) of the kernels of one solve of the 1 M plate --
# the FP64 SpMV, the 16-bit passes and the level-1 pass on the FP32 twin
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pmc2; mkdir -p $O/pmc
cd $R
PROBE="python3 tools/lo_probe.py --reps 4 --solves 1"
timeout -k 10 240 $PROBE > $O/pmc/plain.log 2>&1; echo "plain rc=$?"
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $cnt --output-format csv -d $O/pmc/$cnt -- $PROBE > $O/pmc/$cnt.log 2>&1; echo "$cnt rc=$?"
done
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
find $O/pmc -name "*.csv" -size +1M -delete
grep -A1 -E "k_spmv_win_sym<true|k_spmv_win_lo_sym<[12]|k_spmv_l1_i16|k_stream_read |k_symv_tiles|k_restrict_init|k_prolong_add" $O/pmc_summary.txt | head -60
