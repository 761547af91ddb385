#!/usr/bin/env python3
"""Which allocation decides whether a process gets the fast (0.50 ms) or the slow (0.60 ms) SpMV?  Several model instances
in one process (matrix + index arrays re-allocated), each timed with several freshly allocated (x, y) vector pairs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fcvm-workbench_amd"))
from fcvm_hip import meshgen
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel

case = meshgen.cube_case(55, l=100.0, traction=(20.0, 0.0, 200.0))
xh = np.random.default_rng(0).standard_normal(3 * len(case["nocoord"]))
keep = []
for rep in range(6):
    m = GpuModel(case["elNodes"], case["nocoord"], (210000.0, 0.3, 7.8e-6), case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    c = m.ctx
    c.set_precond("bj")
    m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    ts = []
    vecs = []
    for k in range(4):
        x, y = c.array(xh), c.empty(m.ndof)
        vecs.append((x, y))   # keep them alive so that the next pair gets different memory
        for _ in range(5):
            c.spmv(x, y)
        c.sync()
        t0 = time.perf_counter()
        for _ in range(40):
            c.spmv(x, y)
        c.sync()
        ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print("matrix instance", rep, "| SpMV ms with 4 different (x, y) allocations:", " ".join("%.4f" % t for t in ts), flush=True)
    if rep % 2 == 0:
        keep.append(m)      # every other instance stays alive: the next one cannot reuse its memory
    else:
        c.close()
