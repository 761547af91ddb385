#!/bin/bash
# per-launch durations of the assembly kernels (grid = work items of a length class)
set -o pipefail
O=gpurun_out/r04f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for route in ${ROUTES_LIST:-direct copy}; do
  FCVM_VERBOSE=1 ROUTES=$route rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof_$route -o asm -- python3 $R/tools/sessions/asm_time_r04.py $ASM_ARGS > $R/$O/asm_$route.log 2>&1
  grep "calc_gsm\|work items" $R/$O/asm_$route.log
  f=$(find $R/$O/prof_$route -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "assemble_lds" in r["Kernel_Name"] or "repack" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-12:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print("%-34s grid %8s  lds %7s  start %9.1f us  dur %9.1f us" % (r["Kernel_Name"][:34], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")), (s - t0) / 1e3, (e - s) / 1e3))
P
  rm -rf $R/$O/prof_$route
done
