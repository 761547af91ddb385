"""Round 5, verdict item 3 (continued): the FP64 SpMV of the symmetric window form takes 0.335 ms back to back and 0.385 ms inside PCG on the
same box in the same process.  Which ingredient of the PCG iteration does that?  Phases of one process under rocprofv3 --kernel-trace, separated by
fcvm_membench launches (k_stream_read: the analyser splits on them); every phase launches the SpMV 40 times:
  A  fcvm_spmv back to back (no inner product)            B  fcvm_spmv_dot back to back (the PCG's variant)
  C  spmv_dot, then three vector updates over other 34 MB vectors   D  spmv_dot on a vector that a vector update has just written (x = p of PCG)
  E  spmv_dot, x just written, followed by the two 16-bit passes    F  two elastic solves (the real thing)
  G  as A again
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 spmv_phases_r05.py ; python3 spmv_phases_r05.py --analyse DIR"""
import glob
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))

if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    import csv
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    phases, cur, in_marker = [], [], False
    for r in rows:
        n = r["Kernel_Name"]
        if "k_stream_read" in n:
            if not in_marker:
                phases.append(cur); cur = []
            in_marker = True
            continue
        in_marker = False
        cur.append(r)
    phases.append(cur)
    names = ["set-up", "A spmv back to back", "B spmv_dot back to back", "C spmv_dot + 3 vector updates elsewhere", "D spmv_dot on a vector just written",
             "E spmv_dot (x just written) + two 16-bit passes", "F two elastic solves (PCG)", "G spmv back to back again"]
    for k, ph in enumerate(phases):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ph if "k_spmv_win_sym<" in r["Kernel_Name"]]
        d = [v for v in d if v > 50.0]       # (no-op launches behind a converged solve return at once)
        if not d:
            continue
        d = np.array(d)
        gaps = []
        for i, r in enumerate(ph):
            if "k_spmv_win_sym<" in r["Kernel_Name"] and i > 0:
                gaps.append((int(r["Start_Timestamp"]) - int(ph[i - 1]["End_Timestamp"])) / 1e3)
        print("%-52s n %3d  mean %.1f us  min %.1f  max %.1f  (first five: %s)  gap before launch: median %.1f us" %
              (names[k] if k < len(names) else "phase %d" % k, len(d), d.mean(), d.min(), d.max(), " ".join("%.0f" % v for v in d[:5]), np.median(gaps) if gaps else 0.0))
    sys.exit(0)

from fcvm_hip import examples  # noqa: E402
from fcvm_hip._lib import call  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

case = examples.plate_with_hole(1.45)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
n = m.ndof
rng = np.random.default_rng(0)
x, y, z, a, b = c.array(rng.standard_normal(n)), c.empty(n), c.empty(n), c.array(rng.standard_normal(n)), c.array(rng.standard_normal(n))
out = c.empty(8)
mark = lambda: (c.sync(), c.membench(1 << 26), c.sync())  # noqa: E731
dot = lambda xx, yy: call("fcvm_spmv_dot", c.h, xx.ptr, yy.ptr, out.ptr)  # noqa: E731
for _ in range(5):
    c.spmv(x, y)
N = 40
mark()
for _ in range(N):
    c.spmv(x, y)
mark()
for _ in range(N):
    dot(x, y)
mark()
for _ in range(N):
    dot(x, y)
    c.axpby(1.0, a, 0.5, b, out=z); c.axpby(1.0, z, 0.5, a, out=b); c.axpby(1.0, b, 0.5, z, out=a)
mark()
for _ in range(N):
    c.axpby(1.0, a, 1e-3, x, out=x)
    dot(x, y)
mark()
for _ in range(N):
    c.axpby(1.0, a, 1e-3, x, out=x)
    dot(x, y)
    c.spmv_lo(y, z); c.spmv_lo(z, b)
mark()
f = c.mul(m.fixdof, glv)
c.axpby(1.0, f, 1.0, modf, out=f)
for _ in range(2):
    u, info = c.solve(f, rtol=1e-10)
    tm = c.solve_timing()
    print("solve: %d iterations, in-PCG SpMV by HIP events %.4f ms" % (info["iterations"], tm["spmv_ms"] / max(tm["spmv_timed"], 1)), flush=True)
mark()
for _ in range(N):
    c.spmv(x, y)
mark()
