#!/bin/bash
# round 5, verdict item 6: work items of the assembly into the symmetric form made of whole lane octets, and the lane order of the symmetric form with the
# cut segments clustered (1: inside a length, 2: by wave), against rounds 3-4 (sorted items, cluster 0): calc_gsm and pass times, then PMC WRITE_SIZE /
# FETCH_SIZE of the assembly kernels
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t; mkdir -p $O; rm -f $O/times.txt
cd $R
VARIANTS=${VARIANTS:-sorted,0 sorted,1 octet,1 sorted,2 octet,2}
for v in $VARIANTS; do
  set -- $(echo $v | tr "," " ")
  FCVM_VERBOSE=1 FCVM_ASM_ITEMS=$1 FCVM_SYM_CLUSTER_CUT=$2 timeout -k 10 120 python3 tools/sessions/asm_items_r05.py 2>&1 | grep -E "^\[items|work items" | cut -c1-330 | tee -a $O/times.txt
  [ ${PIPESTATUS[0]} -ne 0 ] && { echo "variant $v failed: stopping"; exit 1; }
done
[ -n "$NOPMC" ] && exit 0
export TMPDIR=/tmp ROUTES=direct FCVM_ASSEMBLE_OVERLAP=0 FCVM_LEVEL1_OVERLAP=0 FCVM_REPACK_OVERLAP=0
PMCVARIANTS=${PMCVARIANTS:-octet,2 sorted,0}
for v in $PMCVARIANTS; do
  set -- $(echo $v | tr "," " ")
  for cnt in WRITE_SIZE FETCH_SIZE; do
    FCVM_ASM_ITEMS=$1 FCVM_SYM_CLUSTER_CUT=$2 timeout -k 10 200 rocprofv3 --pmc $cnt --output-format csv -d $O/pmc_$1_$2/$cnt -- python3 tools/sessions/asm_time_r04.py > $O/pmc_$1_$2_$cnt.log 2>&1; rc=$?; echo "$v $cnt rc=$rc"
    [ $rc -ne 0 ] && exit 1
  done
  cp $O/pmc_$1_$2_WRITE_SIZE.log $O/pmc_$1_$2/plain.log
  python3 tools/pmc_summary.py $O/pmc_$1_$2 > $O/pmc_$1_$2.txt 2>&1
  find $O/pmc_$1_$2 -name "*.csv" -size +1M -delete
  python3 - $O/pmc_$1_$2.txt $v <<'P'
import re, sys
t = open(sys.argv[1]).read()
tot = {}
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    sec = t.split("== %s ==" % cnt)[1].split("== ")[0] if "== %s ==" % cnt in t else ""
    vals = re.findall(r"k_assemble_lds<false, true, \d+>\s+grid\s+\d+\s+n=(\d+)\n\s+%s\s+([\d.]+)" % cnt, sec)
    tot[cnt] = sum(float(x) for _, x in vals)
print(sys.argv[2], "assembly kernels per assembly: FETCH %.3f GB, WRITE %.3f GB" % (tot["FETCH_SIZE"] * 2048 / 1e9, tot["WRITE_SIZE"] * 1024 / 1e9))
P
done
