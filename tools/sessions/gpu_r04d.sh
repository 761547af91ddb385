#!/bin/bash
# round 4: where does the time of the assembly straight into the symmetric form go? kernel trace of both routes
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for route in direct copy; do
  ROUTES=$route rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$route -o asm -- python3 $R/tools/sessions/asm_time_r04.py > $R/$O/asm_$route.log 2>&1
  tail -2 $R/$O/asm_$route.log
done
cd $R
python tools/pmc_summary.py --help > /dev/null 2>&1
for route in direct copy; do
  f=$(find $O/prof_$route -name "*kernel_stats.csv" | head -1)
  echo "== $route: $f"; head -14 "$f" | cut -c1-200
  cp "$f" $O/kernel_stats_$route.csv
done
rm -rf $O/prof_direct $O/prof_copy
