"""Round 5, verdict item 3: the same SpMV on the same matrix takes 0.35 ms with one pair of vectors and 0.39 ms with another (the PCG's own p, q
against a caller's x, y)?  Per model instance: twelve freshly allocated (x, y) pairs, the back-to-back SpMV time with each; then x fixed / y varied
and the other way round."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

case = examples.plate_with_hole(1.45)
keep = []


def t_spmv(c, x, y, n=30):
    for _ in range(3):
        c.spmv(x, y)
    c.sync(); t0 = time.perf_counter()
    for _ in range(n):
        c.spmv(x, y)
    c.sync()
    return (time.perf_counter() - t0) / n * 1e3


for inst in range(int(os.environ.get("VEC_INSTANCES", "3"))):
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    keep.append(m)
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    xh = np.random.default_rng(0).standard_normal(m.ndof)
    xs = [c.array(xh) for _ in range(12)]
    ys = [c.empty(m.ndof) for _ in range(12)]
    pairs = [t_spmv(c, xs[k], ys[k]) for k in range(12)]
    print("instance %d  twelve (x, y) pairs: %s" % (inst, " ".join("%.4f" % v for v in pairs)), flush=True)
    print("instance %d  x0 with every y:     %s" % (inst, " ".join("%.4f" % t_spmv(c, xs[0], ys[k]) for k in range(12))), flush=True)
    print("instance %d  every x with y0:     %s" % (inst, " ".join("%.4f" % t_spmv(c, xs[k], ys[0]) for k in range(12))), flush=True)
    print("instance %d  addresses x %s" % (inst, " ".join("%x" % (v.ptr >> 20) for v in xs)))
    print("instance %d  addresses y %s" % (inst, " ".join("%x" % (v.ptr >> 20) for v in ys)))
    u, info = c.solve(f, rtol=1e-10)
    u, info = c.solve(f, rtol=1e-10)
    tm = c.solve_timing()
    print("instance %d  in PCG %.4f ms" % (inst, tm["spmv_ms"] / max(tm["spmv_timed"], 1)), flush=True)
    import ctypes as C
    from fcvm_hip._lib import call
    from fcvm_hip.device import DArr
    a = (C.c_uint64 * 5)()
    call("fcvm_debug_pcg_vectors", c.h, a)
    P, Q = DArr(c, m.ndof, np.float64, ptr=a[2]), DArr(c, m.ndof, np.float64, ptr=a[3])
    out = c.empty(8)

    def t_dot(xx, yy, n=30):
        for _ in range(3):
            call("fcvm_spmv_dot", c.h, xx.ptr, yy.ptr, out.ptr)
        c.sync(); t0 = time.perf_counter()
        for _ in range(n):
            call("fcvm_spmv_dot", c.h, xx.ptr, yy.ptr, out.ptr)
        c.sync()
        return (time.perf_counter() - t0) / n * 1e3
    print("instance %d  the solver's own p, q (addresses %x %x, r %x z %x partials %x): spmv(p, q) %.4f  spmv(p, y0) %.4f  spmv(x0, q) %.4f  spmv(x0, y0) %.4f | "
          "with the inner product: (p, q) %.4f  (x0, y0) %.4f" %
          (inst, a[2] >> 12, a[3] >> 12, a[0] >> 12, a[1] >> 12, a[4] >> 12, t_spmv(c, P, Q), t_spmv(c, P, ys[0]), t_spmv(c, xs[0], Q), t_spmv(c, xs[0], ys[0]), t_dot(P, Q),
           t_dot(xs[0], ys[0])), flush=True)
