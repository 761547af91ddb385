"""back-to-back time of the FP64 SpMV on the 1 M plate (block-Jacobi context: no hierarchy, no solve) -- for A/B builds of the kernel"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
os.environ["FCVM_PRECOND"] = "bj"
from fcvm_hip import _lib, examples
if os.environ.get("FCVM_AB_LIB"):      # an A/B build of the library (tools/sessions/ab/*.so, not tracked)
    _lib.LIB_PATH = os.path.join(REPO, os.environ["FCVM_AB_LIB"])
from fcvm_hip.device import Context
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
case = examples.plate_with_hole(1.45)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctx=Context(0))
c = m.ctx
m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
x, y = c.array(np.random.default_rng(0).standard_normal(m.ndof)), c.empty(m.ndof)
for rep in range(4):
    for _ in range(5):
        c.spmv(x, y)
    c.sync(); t0 = time.perf_counter()
    for _ in range(50):
        c.spmv(x, y)
    c.sync()
    print("SpMV back to back %.4f ms" % ((time.perf_counter() - t0) / 50 * 1e3), flush=True)
print("stream read %.0f GB/s" % c.membench(3 << 30))
