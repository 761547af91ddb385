#!/bin/bash
# round 3: aggregate size / level-1 steps once more, now that the fine-level passes are cheaper (whole analyses of the 1 M plate)
O=gpurun_out/r03bq
mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date +%T >> $O/heartbeat.txt; done ) &
HB=$!
trap "kill $HB" EXIT
V="lo,lo:env.FCVM_MG_TARGET=100,lo:env.FCVM_MG_TARGET=64,lo:env.FCVM_MG_K1S=2,lo:env.FCVM_MG_K1S=4,lo:env.FCVM_MG_TARGET=100;env.FCVM_MG_K1S=4,lo"
timeout -k 10 700 python3 tools/precond_probe.py --case plate --h 1.45 --variants "$V" --out $O/plate_1m.json > $O/plate_1m.txt 2>&1 || exit 1
python3 - <<'PY'
import json
for r in json.load(open('gpurun_out/r03bq/plate_1m.json')):
    print("%-48s its %.1f  ms/it %.3f  solve %.2f  analysis %.2f" % (r['variant'], r.get('its_per_solve',0), r.get('ms_per_pcg_iteration',0), r.get('solve_s',0), r.get('analysis_s',0)), r.get('error','')[:80])
PY
