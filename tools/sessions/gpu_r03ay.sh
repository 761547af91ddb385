#!/bin/bash
# round 3: would a second fine-level smoothing step pay now that the coarse levels are better?  (FP64 multiplicative form, k0 = 1 / 2)
O=gpurun_out/r03ay
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V="lo,mult1,mult2,mult1:env.FCVM_MG_K1S=4,mult2:env.FCVM_MG_K1S=4,mult2:env.FCVM_MG_K1S=5,mult3:env.FCVM_MG_K1S=4"
timeout -k 10 900 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.txt 2>&1 || exit 1
python3 - <<'PY'
import json
for r in json.load(open('gpurun_out/r03ay/plate_1m.json')):
    print("%-40s its %.1f  ms/it %.3f  solve %.2f  analysis %.2f" % (r['variant'], r.get('its_per_solve',0), r.get('ms_per_pcg_iteration',0), r.get('solve_s',0), r.get('analysis_s',0)), r.get('error','')[:80])
PY
