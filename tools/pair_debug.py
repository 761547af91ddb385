#!/usr/bin/env python3
"""Bitwise comparison of the two-right-hand-side building blocks with their single counterparts (debugging aid)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples, meshgen  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "emb"
if which == "emb":
    case = examples.embankment(refine=2)
elif which == "plate":
    case = examples.plate_with_hole(1.45)
else:
    case = dict(meshgen.cube_case(12, l=10.0, traction=(0, 0, 1.0)), material=(210000.0, 0.3, 7.8e-6))
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]) if "ufaces" in case else Loads())
c = m.ctx
glv, modf, _, _ = m.calc_gsm((0.0, 0.0, -10.0), want_x=False)
rng = np.random.default_rng(5)
fa = c.mul(m.fixdof, glv)
fb = c.array(m.fixdof_h * rng.standard_normal(m.ndof))
e = lambda: c.empty(m.ndof)  # noqa: E731


def diff(tag, a, b):
    a, b = a.get(), b.get()
    d = np.abs(a - b)
    print("%-28s equal %s  max |diff| %.3e  (max |value| %.3e, %d entries differ)" % (tag, np.array_equal(a, b), d.max(), np.abs(b).max(), int((d > 0).sum())))


print(c.spmv_info(), c.precond_info()["cycle_form"])
ya, yb = c.spmv2(fa, fb, e(), e())
diff("spmv2 A", ya, c.spmv(fa, e())); diff("spmv2 B", yb, c.spmv(fb, e()))
ya, yb = c.spmv2(fb, fb, e(), e())
diff("spmv2 (B,B): A vs B", ya, yb)
za, zb = c.mg_apply2(fa, fb, e(), e())
diff("mg_apply2 A", za, c.mg_apply(fa, e())); diff("mg_apply2 B", zb, c.mg_apply(fb, e()))
za, zb = c.mg_apply2(fb, fb, e(), e())
diff("mg_apply2 (B,B): A vs B", za, zb)
z1, z2 = c.mg_apply(fb, e()), c.mg_apply(fb, e())
diff("mg_apply twice", z1, z2)
ua, ia = c.solve(fa, rtol=1e-11)
ub, ib = c.solve(fb, rtol=1e-11)
va, vb, ja, jb = c.solve2(fa, fb, rtol=(1e-11, 1e-11))
print(ia, ib, ja, jb)
diff("solve2 A", va, ua); diff("solve2 B", vb, ub)
if len(sys.argv) > 2 and sys.argv[2] == "time":
    import time
    for _ in range(2):
        c.sync(); t0 = time.perf_counter()
        c.solve(fa, rtol=1e-10); c.solve(fb, rtol=1e-10)
        c.sync(); t1 = time.perf_counter()
        c.solve2(fa, fb, rtol=(1e-10, 1e-10))
        c.sync(); t2 = time.perf_counter()
        print("two solves one after the other %.1f ms, in lockstep %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
