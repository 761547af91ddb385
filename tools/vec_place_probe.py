#!/usr/bin/env python3
"""Does the address of the x / y vectors change the SpMV time?  (bench session r02l: 0.62 vs 0.55 ms between candidate pairs
of the placement tuner, identical value arrays.)  One model, K freshly allocated (x, y) pairs plus sub-ranges of one large
allocation at chosen byte offsets; interleaved rounds, median per pair, addresses printed.

  python3 tools/vec_place_probe.py --cells 55 --pairs 6
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import meshgen  # noqa: E402
from fcvm_hip.device import Context, DArr  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=55)
ap.add_argument("--pairs", type=int, default=6)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()

case = meshgen.cube_case(a.cells, l=100.0, traction=(20.0, 0.0, 200.0))
ctx = Context(0)
ctx.set_precond("bj")
m = GpuModel(case["elNodes"], case["nocoord"], (210000.0, 0.3, 7.8e-6), case["fix"], Loads(), ctx=ctx)
m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
n = m.ndof
xh = np.random.default_rng(0).standard_normal(n)
pairs = []
for k in range(a.pairs):
    x, y = ctx.array(xh), ctx.empty(n)
    pairs.append(("fresh %d" % k, x, y))
# one large allocation, x at its start, y at chosen distances
big = ctx.empty(16 * (n + (1 << 20)))
for name, off_x, off_y in (("same alloc, y right after x", 0, (n * 8 + 255) // 256 * 256), ("same alloc, y at +64 MiB", 0, 64 << 20), ("same alloc, y at +64 MiB + 4 KiB", 0, (64 << 20) + 4096),
                           ("same alloc, x at +2 MiB+1 KiB, y at +128 MiB + 37 KiB", (2 << 20) + 1024, (128 << 20) + 37 * 1024),
                           ("same alloc, y BEFORE x", 256 << 20, 0)):
    assert max(off_x, off_y) + n * 8 <= big.nbytes and off_x % 16 == 0 and off_y % 16 == 0
    x = DArr(ctx, n, np.float64, ptr=big.ptr + off_x)
    y = DArr(ctx, n, np.float64, ptr=big.ptr + off_y)
    x.set(xh)
    pairs.append((name, x, y))
times = {p[0]: [] for p in pairs}
for r in range(a.rounds + 1):
    for name, x, y in pairs:
        ctx.spmv(x, y)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            ctx.spmv(x, y)
        ctx.sync()
        if r:
            times[name].append((time.perf_counter() - t0) / a.reps * 1e3)
for name, x, y in pairs:
    t = np.array(times[name])
    print("%-58s x %#x (mod 2 MiB %#8x)  y %#x (mod 2 MiB %#8x)  y - x %+d MiB  median %.4f ms  min %.4f  max %.4f" %
          (name, x.ptr, x.ptr % (2 << 20), y.ptr, y.ptr % (2 << 20), (y.ptr - x.ptr) >> 20, np.median(t), t.min(), t.max()))
