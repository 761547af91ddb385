"""Round 5, verdict item 3: inside PCG the FP64 SpMV takes ~0.38 ms (kernel time) in every process, back to back 0.335-0.38 depending on the
process.  Is it the near-idle microseconds in front of it (a PCG iteration has single-workgroup kernels: k_pcg_scalars, k_sum_final)?  Per model
instance, under rocprofv3 --kernel-trace, phases separated by fcvm_membench launches:
  A  40 x spmv back to back      B  40 x [one tiny kernel pair (dot of 8 numbers), spmv]      C  40 x [ten tiny kernel pairs, spmv]
  D  40 x [three 34 MB vector updates, spmv]      E  two elastic solves
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 spmv_gaps_r05.py ; python3 spmv_gaps_r05.py --analyse DIR"""
import glob
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
NINST = int(os.environ.get("GAP_INSTANCES", "3"))
NAMES = ["A back to back", "B one tiny kernel pair before each", "C ten tiny kernel pairs before each", "D three 34 MB vector updates before each", "E two elastic solves (PCG)"]

if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    import csv
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    phases, cur, in_marker = [], [], False
    for r in rows:
        if "k_stream_read" in r["Kernel_Name"]:
            if not in_marker:
                phases.append(cur); cur = []
            in_marker = True
            continue
        in_marker = False
        cur.append(r)
    phases.append(cur)
    k = 0
    for ph in phases:
        d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ph if "k_spmv_win_sym<" in r["Kernel_Name"]])
        d = d[d > 50.0]
        if len(d) < 30:
            continue
        print("instance %d  %-44s n %3d  mean %.1f us  min %.1f  max %.1f" % (k // len(NAMES), NAMES[k % len(NAMES)], len(d), d.mean(), d.min(), d.max()))
        k += 1
    sys.exit(0)

from fcvm_hip import examples  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

case = examples.plate_with_hole(1.45)
keep = []
for inst in range(NINST):
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    keep.append(m)
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    n = m.ndof
    rng = np.random.default_rng(0)
    x, y, z, a, b = c.array(rng.standard_normal(n)), c.empty(n), c.empty(n), c.array(rng.standard_normal(n)), c.array(rng.standard_normal(n))
    s8 = c.array(np.ones(8))
    mark = lambda: (c.sync(), c.membench(1 << 26), c.sync())  # noqa: E731
    for _ in range(5):
        c.spmv(x, y)
    N = 40
    mark()
    for _ in range(N):
        c.spmv(x, y)
    mark()
    for _ in range(N):
        c.dot(s8, s8) if False else c.axpby(1.0, s8, 0.0, None, out=s8)
        c.spmv(x, y)
    mark()
    for _ in range(N):
        for _k in range(10):
            c.axpby(1.0, s8, 0.0, None, out=s8)
        c.spmv(x, y)
    mark()
    for _ in range(N):
        c.axpby(1.0, a, 0.5, b, out=z); c.axpby(1.0, z, 0.5, a, out=b); c.axpby(1.0, b, 0.5, z, out=a)
        c.spmv(x, y)
    mark()
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    for _ in range(2):
        u, info = c.solve(f, rtol=1e-10)
    tm = c.solve_timing()
    print("instance %d: solve %d iterations, in-PCG SpMV by HIP events %.4f ms" % (inst, info["iterations"], tm["spmv_ms"] / max(tm["spmv_timed"], 1)), flush=True)
    mark()
