#!/bin/bash
O=gpurun_out/r02i
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "assemble or tangent or cube_analysis or embankment_analysis or bitwise or gnly_updated or buckling_factors" > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
FCVM_ASSEMBLE_OVERLAP=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_seq -- python3 tools/spmv_probe.py --reps 2 --pcg 0 > $O/trace_seq.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_ovl -- python3 tools/spmv_probe.py --reps 2 --pcg 0 > $O/trace_ovl.log 2>&1
for g in "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $g | cut -d' ' -f1)
  FCVM_ASSEMBLE_OVERLAP=0 timeout -k 10 200 rocprofv3 --pmc $g --output-format csv -d $O/pmc_$n -- python3 tools/spmv_probe.py --reps 2 --pcg 0 > $O/pmc_$n.log 2>&1
done
echo done
