#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 rocpd database between two launches of a marker kernel:
   python3 tools/rocpd_timeline.py out.db k_assemble_lds [occurrence] -- prints the dispatches from that launch of the marker to the next
   k_spmv_win launch after it (name, start offset, duration, stream), i.e. one tangent refresh of the GNLY loop."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_assemble_lds"
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 3
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
kv = [v for v in views if v == "kernels"] or [v for v in views if "kernel" in v.lower()]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kv[0])]
print("# view", kv[0], cols, file=sys.stderr)
name_c = "name" if "name" in cols else [c for c in cols if "name" in c][0]
st_c = "start" if "start" in cols else [c for c in cols if "start" in c][0]
en_c = "end" if "end" in cols else [c for c in cols if "end" in c][0]
q_c = [c for c in cols if c in ("stream", "stream_id", "queue", "queue_id")]
rows = list(db.execute("select %s, %s, %s%s from %s order by %s" % (name_c, st_c, en_c, ", " + q_c[0] if q_c else "", kv[0], st_c)))
idx = [i for i, r in enumerate(rows) if marker in r[0]]
# group consecutive marker launches (the buckets of one assembly) into occurrences
groups, last = [], None
for i in idx:
    if last is None or rows[i][1] - rows[last][1] > 20e6:
        groups.append(i)
    last = i
i0 = groups[min(occ, len(groups) - 1)]
t0 = rows[i0][1]
n_spmv = 0
for r in rows[max(i0 - 3, 0):]:
    if r[1] - t0 > 20e6:
        break
    print("%9.1f us  +%8.1f us  %-8s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if len(r) > 3 else "", r[0][:90]))
