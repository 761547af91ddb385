"""smoother bounds from 15 / 40 / 100 power iterations and what they do to the PCG on the 1 M plate"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.device import Context
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
name = sys.argv[1] if len(sys.argv) > 1 else "plate"
case = examples.plate_with_hole(float(os.environ.get("PLATE_H", "1.45"))) if name == "plate" else examples.embankment(refine=3)
grav = (0.0, 0.0, 0.0) if name == "plate" else (0.0, 0.0, -10.0)
for iters in (15, 40, 100, 300):
    os.environ["FCVM_MG_LMAX_ITERS"] = str(iters)
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                 Loads(ufaces=case["ufaces"], uloads=case["uloads"]) if "ufaces" in case else Loads(), ctx=Context(0))
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm(grav, want_x=False)
    f = c.mul(m.fixdof, glv)
    rng = np.random.default_rng(2)
    g = c.array(m.fixdof_h * rng.standard_normal(m.ndof))
    its = []
    for rhs in (f, g):
        c.sync(); t0 = time.perf_counter()
        u, info = c.solve(rhs, rtol=1e-10)
        c.sync(); its.append((info["iterations"], (time.perf_counter() - t0) * 1e3))
    pi = c.precond_info()
    print("%s: %3d power iterations: lmax0 %.5f lmax1 %.5f | PCG iterations / ms: load vector %d / %.1f, random vector %d / %.1f" %
          (name, iters, pi["lmax0"], pi["lmax1"], its[0][0], its[0][1], its[1][0], its[1][1]), flush=True)
    c.close()
