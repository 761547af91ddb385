#!/usr/bin/env python3
"""SpMV probe for profiling / tuning: assembles the synthetic cube and launches the SpMV kernel
`reps` times (plus one short PCG solve so that the dot-fused variant shows up as well).
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/spmv_probe.py --cells 55"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import meshgen  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=55)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--pcg", type=int, default=25, help="PCG iterations of the extra solve (0 = skip)")
ap.add_argument("--membench", action="store_true", help="also run the plain streaming-read kernel (3 GiB)")
a = ap.parse_args()
case = meshgen.cube_case(a.cells, l=100.0, traction=(20.0, 0.0, 200.0))
m = GpuModel(case["elNodes"], case["nocoord"], (210000.0, 0.3, 7.8e-6), case["fix"],
             Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
glv, modf, V, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
info = c.pattern_info()
x = c.array(np.random.default_rng(0).standard_normal(m.ndof))
y = c.empty(m.ndof)
for _ in range(3):
    c.spmv(x, y)
c.sync()
t0 = time.perf_counter()
for _ in range(a.reps):
    c.spmv(x, y)
c.sync()
ms = (time.perf_counter() - t0) / a.reps * 1e3
alg = info["nnz_blocks"] * 76 + 2 * info["ndof"] * 8
print("elements %d ndof %d nnzb %d padded_slots %d  spmv %.4f ms  %.1f GB/s algorithmic (%.1f GB/s incl. padding)" %
      (m.ne, m.ndof, info["nnz_blocks"], info["padded_slots"], ms, alg / ms / 1e6,
       (info["matrix_bytes"] + 2 * info["ndof"] * 8) / ms / 1e6))
w = c.spmv_info()
print("spmv kernel:", ("window format, %d groups, windows mean %d / max %d nodes, %d B LDS, %.4f GB streamed per launch" %
                       (w["groups"], w["mean_window"], w["max_window"], w["lds_bytes"], w["bytes"] / 1e9)) if w["window_format"] else "sliced-ELL")
if a.membench:
    print("stream read %.0f GB/s" % c.membench(3 << 30))
if a.pcg:
    f = c.mul(m.fixdof, glv)
    try:
        c.solve(f, rtol=1e-30, maxit=a.pcg)
    except Exception as e:  # maxit reached is expected here
        print("pcg:", str(e)[:80])
    print("pcg spmv avg ms", c.solve_timing()["spmv_ms"] / a.pcg)
