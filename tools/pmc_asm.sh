#!/bin/bash
# PMC passes over the assembly straight into the symmetric window form on the 1 M plate (k_assemble_lds<.., SYM>, seven length
# classes = seven launches per assembly): one small counter group per run (rocprofv3 --pmc only, no trace domains).
# usage: tools/pmc_asm.sh <outdir> [tangent]
O=$1; shift
mkdir -p $O
export TMPDIR=/tmp
PROBE="python3 tools/sessions/asm_time_r04.py $*"
export ROUTES=direct FCVM_ASSEMBLE_OVERLAP=0 FCVM_LEVEL1_OVERLAP=0 FCVM_REPACK_OVERLAP=0   # one stream: counters are attributed per dispatch
run() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- $PROBE > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" >> $O/passes.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name hung: stopping" >> $O/passes.txt; exit 1; fi
}
timeout -k 10 240 $PROBE > $O/plain.log 2>&1 || exit 1
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq3 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $O > $O/summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
