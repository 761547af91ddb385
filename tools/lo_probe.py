#!/usr/bin/env python3
"""Kernel-level probe of the multiplicative cycle with the 16-bit smoother copy: assembles the 1 M-element plate once and applies the
preconditioner / solves repeatedly (run under rocprofv3 --kernel-trace --stats for per-kernel times).
  python tools/lo_probe.py [--h 1.45] [--form 2] [--reps 30]"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))

from fcvm_hip import examples  # noqa: E402
from fcvm_hip.device import Context  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--h", type=float, default=1.45)
ap.add_argument("--form", type=int, default=2)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--solves", type=int, default=3)
a = ap.parse_args()
case = examples.plate_with_hole(a.h)
ctx = Context(0)
ctx.set_mg_cycle(a.form)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctx=ctx)
c = m.ctx
glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
f = c.mul(m.fixdof, glv)
z = c.empty(m.ndof)
for _ in range(3):
    c.mg_apply(f, z)
c.sync()
t0 = time.perf_counter()
for _ in range(a.reps):
    c.mg_apply(f, z)
c.sync()
print("cycle form %d: %.3f ms per preconditioner application (%d tet10)" % (a.form, (time.perf_counter() - t0) / a.reps * 1e3, m.ne))
for _ in range(a.solves):
    t0 = time.perf_counter()
    u, info = c.solve(f, rtol=1e-10)
    c.sync()
    dt = time.perf_counter() - t0
    print("solve: %d iterations, %.1f ms, %.3f ms per iteration" % (info["iterations"], dt * 1e3, dt * 1e3 / max(info["iterations"], 1)))
x, y = c.array(np.random.default_rng(0).standard_normal(m.ndof)), c.empty(m.ndof)
for _ in range(3):
    c.spmv(x, y)
c.sync()
t0 = time.perf_counter()
for _ in range(12):
    c.spmv(x, y)       # k_spmv_win<false>: every launch live (counter passes: traffic per launch without the PCG's no-op launches)
c.sync()
info = c.pattern_info()
print("plain SpMV back to back: %.4f ms; nnz blocks %d, dof %d, window format blocks %d" %
      ((time.perf_counter() - t0) / 12 * 1e3, info["nnz_blocks"], info["ndof"], c.spmv_info()["blocks"]))
print("stream read %.0f GB/s" % c.membench(3 << 30))
