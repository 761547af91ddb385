"""How fast is the 16-bit window-format pass on a matrix the size of the level-1 operator of the 1 M plate (176 k nodes)?  The fine matrix of the plate
at h = 2.9 has 180 k nodes: its 16-bit pass against the 24 us of k_spmv_l1_i16 on the level-1 matrix (70 MB) -- is a window-format level 1 worth building?"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
for h in (2.9, 2.3):
    case = examples.plate_with_hole(h)
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    c = m.ctx
    m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    x, y = c.array(np.random.default_rng(0).standard_normal(m.ndof)), c.empty(m.ndof)
    si = c.spmv_sym_info()
    for name, fn in (("FP64 SpMV", lambda: c.spmv(x, y)), ("16-bit pass", lambda: c.spmv_lo(x, y))):
        for _ in range(10):
            fn()
        c.sync(); t0 = time.perf_counter()
        for _ in range(200):
            fn()
        c.sync()
        print("h %.2f: %d tet10, %d nodes, %d stored blocks (%.1f MB in 16 bit): %s %.1f us" % (h, m.ne, m.nn, si["blocks"], si["blocks"] * 24e-6, name, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
