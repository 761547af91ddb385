"""Is the vertex-patch cycle reproducible?  The Embankment case of the test, N times in one process on memory dirtied with 0xFF bytes first
(a fresh process sees zero pages from the driver; a long test run does not): iterations, lambda_max and the solution's checksum per run."""
import os, sys, hashlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
os.environ["FCVM_MG_PATCH"] = "1"
from fcvm_hip import examples
from fcvm_hip.device import Context
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
import ctypes as C
from fcvm_hip._lib import call
c0 = Context(0)
os.environ["FCVM_POISON"] = "1"
ptrs = []
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):      # GB
    p = C.c_void_p()
    call("fcvm_malloc", c0.h, C.c_uint64(1 << 30), C.byref(p))
    ptrs.append(p)
os.environ["FCVM_POISON"] = "0"
for p in ptrs:
    call("fcvm_free", c0.h, p)
case = examples.embankment(refine=1)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(), ctx=Context(0))
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm((0.0, 0.0, -10.0), want_x=False)
    f = c.axpby(1.0, c.mul(m.fixdof, glv), 1.0, modf)
    try:
        u, info = c.solve(f, rtol=1e-10)
        print("run %2d: %d iterations, lmax %.6f, weight check %.4f, checksum %s" % (k, info["iterations"], c.patch_info()["lmax"], 2.0 / (1.02 * 1.0625),
              hashlib.md5(u.get().tobytes()).hexdigest()[:12]), flush=True)
    except Exception as e:
        print("run %2d: FAILED %s (lmax %.6f)" % (k, str(e)[:120], c.patch_info()["lmax"]), flush=True)
    c.close()
