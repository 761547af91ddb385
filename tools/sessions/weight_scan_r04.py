"""damping of the fine-level Jacobi step of cycle form 2 (FCVM_MG_ADD_W scales c0; c0 * lambda_max = 1.86 at 1.0) and level-1
Chebyshev interval ratio against PCG iterations on the 1 M plate / the 337k Embankment"""
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
settings = [("FCVM_MG_ADD_W", w) for w in ("0.8", "0.9", "0.95", "1.0", "1.04", "1.07")] + [("FCVM_MG_ALPHA1S", a) for a in ("6", "9", "16", "24")]
for key, val in settings:
    os.environ.pop("FCVM_MG_ADD_W", None); os.environ.pop("FCVM_MG_ALPHA1S", None)
    os.environ[key] = val
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                 Loads(ufaces=case["ufaces"], uloads=case["uloads"]) if "ufaces" in case else Loads(), ctx=Context(0))
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm(grav, want_x=False)
    f = c.mul(m.fixdof, glv)
    g = c.array(m.fixdof_h * np.random.default_rng(2).standard_normal(m.ndof))
    its = []
    for rhs in (f, g):
        c.sync(); t0 = time.perf_counter()
        u, info = c.solve(rhs, rtol=1e-10)
        c.sync(); its.append((info["iterations"], (time.perf_counter() - t0) * 1e3))
    print("%s %s=%s: PCG iterations / ms: load vector %d / %.1f, random vector %d / %.1f" % (name, key, val, its[0][0], its[0][1], its[1][0], its[1][1]), flush=True)
    c.close()
