"""time of a pair of solves (fcvm_solve2) against two single solves on the 1 M plate"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
case = examples.plate_with_hole(1.45)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
fa = c.mul(m.fixdof, glv)
fb = c.array(m.fixdof_h * np.random.default_rng(5).standard_normal(m.ndof))
for rep in range(3):
    c.sync(); t0 = time.perf_counter()
    ua, ia = c.solve(fa, rtol=1e-10); ub, ib = c.solve(fb, rtol=1e-10)
    c.sync(); t1 = time.perf_counter()
    va, vb, ja, jb = c.solve2(fa, fb, rtol=(1e-10, 1e-10))
    c.sync(); t2 = time.perf_counter()
    print("two single solves %.1f ms (%d + %d iterations), pair %.1f ms; bitwise equal: %s" % ((t1 - t0) * 1e3, ia["iterations"], ib["iterations"], (t2 - t1) * 1e3,
          np.array_equal(ua.get(), va.get()) and np.array_equal(ub.get(), vb.get())))
