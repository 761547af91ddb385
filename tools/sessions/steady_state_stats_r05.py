"""Per-PCG-iteration kernel table from a rocprofv3 kernel trace (csv): every dispatch between the a-th and the b-th launch of a marker kernel that
runs once per iteration (k_dist_cg_update for the host-sequenced multi-GPU cycle, k_pcg_update_p for the library's own loop), summed by kernel
and divided by the number of iterations in between.   python3 steady_state_stats_r05.py DIR MARKER [a b]"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
b = int(sys.argv[4]) if len(sys.argv) > 4 else len(idx) - 2
i0, i1 = idx[a], idx[b]
n = b - a
tot = defaultdict(lambda: [0, 0.0])
for r in rows[i0:i1]:
    k = r["Kernel_Name"].split("(")[0][:110]
    tot[k][0] += 1
    tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
wall = (int(rows[i1]["Start_Timestamp"]) - int(rows[i0]["Start_Timestamp"])) / 1e3
ksum = sum(v[1] for v in tot.values())
print("%d iterations (launches %d..%d of %s): wall %.1f us per iteration, kernels %.1f us per iteration, %d launches per iteration" %
      (n, a, b, marker, wall / n, ksum / n, sum(v[0] for v in tot.values()) / n))
print("%-112s %9s %10s %9s" % ("kernel", "per iter.", "us / iter.", "avg us"))
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-112s %9.2f %10.1f %9.2f" % (k, c / n, t / n, t / c))
torch_rows = [k for k in tot if "at::native" in k]
print("at::native (torch) kernels inside these iterations: %d" % len(torch_rows))
