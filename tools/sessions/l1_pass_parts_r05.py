"""What does the epilogue of the level-1 pass cost?  k_spmv_l1_i16 as a plain product (fcvm_mg_l1_spmv) against the pass with the Chebyshev step as epilogue
(fcvm_mg_l1_pass), back to back, 1 M plate."""
import os, sys, time, ctypes as C
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip._lib import call
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
case = examples.plate_with_hole(1.45)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
nc = c.precond_info()["coarse_nodes"]
rng = np.random.default_rng(0)
din, dout, x, r, y = (c.array(rng.standard_normal(3 * nc)) for _ in range(5))
def timed(fn, n=300):
    for _ in range(20): fn()
    c.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    c.sync(); return (time.perf_counter() - t0) / n * 1e6
t_plain = timed(lambda: call("fcvm_mg_l1_spmv", c.h, din.ptr, y.ptr))
t_pass = timed(lambda: call("fcvm_mg_l1_pass", c.h, din.ptr, dout.ptr, x.ptr, r.ptr, None, None, None, C.c_double(0.3), C.c_double(0.01), 1, 0))
t_noupd = timed(lambda: call("fcvm_mg_l1_pass", c.h, din.ptr, dout.ptr, x.ptr, r.ptr, None, None, None, C.c_double(0.0), C.c_double(0.0), 0, 0))
print("level 1: %d nodes; plain product %.1f us, pass with the full step %.1f us, pass without a new direction %.1f us (back to back, no twin: a caller's own vectors)" % (nc, t_plain, t_pass, t_noupd))
