#!/bin/bash
# round-2 GPU session ZN: what the window-format SpMV's wavefronts spend their time on (SQ counters, LDS conflicts), final code
O=gpurun_out/r02zn
mkdir -p $O/pmc
export TMPDIR=/tmp
PROBE="python3 tools/spmv_probe.py --reps 6 --pcg 0 --membench"
run() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc/$name -- $PROBE > $O/pmc_$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name hung: stopping"; exit 1; fi
}
timeout -k 10 240 python3 tools/spmv_probe.py --reps 30 --pcg 0 --membench > $O/pmc/plain.log 2>&1 || exit 1
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run ta TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
run grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
grep -A5 "k_spmv_win\|^== \|^elements\|^spmv kernel\|^stream" $O/pmc_summary.txt | grep -v "k_spmv<\|k_assemble\|k_repack\|k_stream" | cut -c1-120
