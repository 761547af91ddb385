#!/usr/bin/env python3
"""A/B of SpMV storage orders in ONE process (interleaved rounds, median / min per variant): the same mesh is set up in
several contexts that differ only in environment knobs read at context creation (FCVM_SPMV_VARIANT, FCVM_VALS_TRIES, ...).

  python3 tools/spmv_ab.py --cells 55 --variants "FCVM_SPMV_VARIANT=4" "FCVM_SPMV_VARIANT=0" --rounds 12 --reps 20
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import meshgen  # noqa: E402
from fcvm_hip.device import Context  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=55)
ap.add_argument("--mesh", default="cube", help="cube | embankment3 | plate")
ap.add_argument("--variants", nargs="+", default=["FCVM_SPMV_VARIANT=4", "FCVM_SPMV_VARIANT=0"])
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()

if a.mesh == "cube":
    case = meshgen.cube_case(a.cells, l=100.0, traction=(20.0, 0.0, 200.0))
    case["material"] = (210000.0, 0.3, 7.8e-6)
elif a.mesh == "embankment3":
    from fcvm_hip import examples
    case = examples.embankment(refine=3)
else:
    from fcvm_hip import examples
    case = examples.plate_with_hole(1.45)

models = []
for v in a.variants:
    saved = {}
    for kv in v.split(","):
        k, val = kv.split("=")
        saved[k] = os.environ.get(k)
        os.environ[k] = val
    ctx = Context(0)
    ctx.set_precond("bj")   # the matrix only
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(), ctx=ctx)
    m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    for k, old in saved.items():
        if old is None:
            del os.environ[k]
        else:
            os.environ[k] = old
    xh = np.random.default_rng(0).standard_normal(m.ndof)
    x = ctx.array(xh)
    y = ctx.empty(m.ndof)
    models.append((v, m, x, y, False))

info = models[0][1].ctx.pattern_info()
alg = info["nnz_blocks"] * 76 + 2 * info["ndof"] * 8
ref = None
times = {v: [] for v, *_ in models}
for r in range(a.rounds + 1):
    for v, m, x, y, pad in models:
        c = m.ctx
        f = c.spmv
        f(x, y)
        c.sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            f(x, y)
        c.sync()
        if r > 0:
            times[v].append((time.perf_counter() - t0) / a.reps * 1e3)
for v, m, x, y, pad in models:
    yy = y.get()
    if ref is None:
        ref = yy
    t = np.array(times[v])
    print("%-40s median %.4f ms  min %.4f  max %.4f  -> %.0f GB/s algorithmic (%.3f of 8 TB/s)  max|y - y0| / max|y0| = %.1e" %
          (v, np.median(t), t.min(), t.max(), alg / np.median(t) / 1e6, alg / np.median(t) / 1e6 / 8000.0,
           np.abs(yy - ref).max() / np.abs(ref).max()))
print("elements %d ndof %d nnzb %d algorithmic bytes %.4f GB, stored matrix %.4f GB" %
      (models[0][1].ne, models[0][1].ndof, info["nnz_blocks"], alg / 1e9, info["matrix_bytes"] / 1e9))
