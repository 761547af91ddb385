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
NINST = int(os.environ.get("GAP_INSTANCES", "2"))
NAMES = ["set-up (Lanczos etc.)", "A spmv back to back", "B spmv_dot back to back", "C spmv_dot, x = the solver's last search direction-like vector (a solution)", "(one solve)", "D two solves, HIP events around every SpMV (default)", "(set-up of the second context)", "E second context: spmv back to back", "F second context (FCVM_PCG_EVENTS=0): two solves, no events"]

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
        if len(d) < 20:
            continue
        print("instance %d  %-44s n %3d  mean %.1f us  min %.1f  max %.1f" % (k // len(NAMES), NAMES[k % len(NAMES)], len(d), d.mean(), d.min(), d.max()))
        k += 1
    sys.exit(0)

from fcvm_hip import examples  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

from fcvm_hip._lib import call  # noqa: E402
case = examples.plate_with_hole(1.45)
keep = []
for inst in range(NINST):
    # two contexts per instance: the default one and one created with FCVM_PCG_EVENTS=0 (no events around the PCG's SpMV launches)
    os.environ.pop("FCVM_PCG_EVENTS", None)
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    keep.append(m)
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    n = m.ndof
    rng = np.random.default_rng(0)
    x, y = c.array(rng.standard_normal(n)), c.empty(n)
    out = c.empty(8)
    mark = lambda: (c.sync(), c.membench(1 << 26), c.sync())  # noqa: E731
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    for _ in range(5):
        c.spmv(x, y)
    N = 40
    mark()
    for _ in range(N):
        c.spmv(x, y)
    mark()
    for _ in range(N):
        call("fcvm_spmv_dot", c.h, x.ptr, y.ptr, out.ptr)
    mark()
    u, info = c.solve(f, rtol=1e-10)
    mark()      # (the solve's launches: a phase of its own, skipped by the analyser's labels below? no: counted -- see NAMES)
    for _ in range(N):
        call("fcvm_spmv_dot", c.h, u.ptr, y.ptr, out.ptr)
    mark()
    for _ in range(2):
        u, info = c.solve(f, rtol=1e-10)
    tm = c.solve_timing()
    print("instance %d: solve %d iterations, in-PCG SpMV by HIP events %.4f ms" % (inst, info["iterations"], tm["spmv_ms"] / max(tm["spmv_timed"], 1)), flush=True)
    mark()
    os.environ["FCVM_PCG_EVENTS"] = "0"
    m2 = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    keep.append(m2)
    c2 = m2.ctx
    glv2, modf2, _, _ = m2.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    f2 = c2.mul(m2.fixdof, glv2)
    c2.axpby(1.0, f2, 1.0, modf2, out=f2)
    x2, y2 = c2.array(rng.standard_normal(n)), c2.empty(n)
    mark2 = lambda: (c2.sync(), c.sync(), c.membench(1 << 26), c.sync())  # noqa: E731  (the marker runs on the first context's stream)
    mark2()
    for _ in range(N):
        c2.spmv(x2, y2)
    mark2()
    for _ in range(2):
        u2, info2 = c2.solve(f2, rtol=1e-10)
    mark2()
