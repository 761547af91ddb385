"""does the placement of the symmetric form's value array decide the FP64 SpMV's time?  Several model instances of the 1 M plate in
one process (every other one kept alive so that the next gets other memory), SpMV back to back and stream-read rate per instance"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.device import Context
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
case = examples.plate_with_hole(1.45)
xh = np.random.default_rng(0).standard_normal(3 * len(case["nocoord"]))
keep = []
for rep in range(6):
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctx=Context(0))
    c = m.ctx
    c.set_precond("bj")
    m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    ts = []
    vecs = []
    for k in range(3):
        x, y = c.array(xh), c.empty(m.ndof)
        vecs.append((x, y))
        for _ in range(5):
            c.spmv(x, y)
        c.sync()
        t0 = time.perf_counter()
        for _ in range(40):
            c.spmv(x, y)
        c.sync()
        ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print("matrix instance", rep, "| SpMV ms with 3 different (x, y) allocations:", " ".join("%.4f" % t for t in ts), "| stream read %.0f GB/s" % c.membench(3 << 30), flush=True)
    if rep % 2 == 0:
        keep.append(m)
    else:
        c.close()
