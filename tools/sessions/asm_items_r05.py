"""Round 5, verdict item 6: calc_gsm (elastic, tangent) and the three fine-level passes on the 1 M plate for one setting of the assembly's work items
(FCVM_ASM_ITEMS = octet | sorted) and of the symmetric form's lane order (FCVM_SYM_CLUSTER_CUT = 0 | 1 | 2), read from the environment."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
case = examples.plate_with_hole(1.45)
ne, nn = len(case["elNodes"]), len(case["nocoord"])
rng = np.random.default_rng(1)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
tag = "[items %s, cluster %s]" % (os.environ.get("FCVM_ASM_ITEMS", "octet"), os.environ.get("FCVM_SYM_CLUSTER_CUT", "1"))
for name, kw in (("elastic", {}), ("tangent", dict(disp=c.array(rng.standard_normal(3 * nn) * 1e-3), sig=c.array(rng.standard_normal(24 * ne) * 50.0),
                                                   pgp=c.array((rng.random(4 * ne) < 0.3).astype(np.uint8), np.uint8), Et_E=0.0))):
    ts = []
    for rep in range(6):
        c.sync(); t0 = time.perf_counter()
        m.calc_gsm((0.0, 0.0, 0.0), want_x=False, **kw)
        c.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s %s calc_gsm %s ms" % (tag, name, " ".join("%.2f" % t for t in ts)), flush=True)
glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
x, y = c.array(rng.standard_normal(m.ndof)), c.empty(m.ndof)
for name, fn in (("FP64 SpMV", lambda: c.spmv(x, y)), ("16-bit pass", lambda: c.spmv_lo(x, y))):
    for _ in range(5):
        fn()
    c.sync(); t0 = time.perf_counter()
    for _ in range(50):
        fn()
    c.sync()
    print("%s %s %.4f ms" % (tag, name, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
f = c.mul(m.fixdof, glv)
c.axpby(1.0, f, 1.0, modf, out=f)
u, info = c.solve(f, rtol=1e-10)
u, info = c.solve(f, rtol=1e-10)
tm = c.solve_timing()
si = c.spmv_sym_info()
print("%s solve %d iterations %.1f ms, in-PCG SpMV %.4f ms, stored blocks %d" % (tag, info["iterations"], tm["total_ms"], tm["spmv_ms"] / max(tm["spmv_timed"], 1), si["blocks"]), flush=True)
