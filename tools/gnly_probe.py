#!/usr/bin/env python3
"""Cost of a GNLY Newton iteration (tangent re-assembly every iteration) on the synthetic cube: where does the time go?
  python3 tools/gnly_probe.py --cells 55"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from fcvm_hip import meshgen  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel, RiksDriver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=55)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
case = meshgen.cube_case(a.cells, l=100.0, traction=bench.TRACTION)
m = GpuModel(case["elNodes"], case["nocoord"], bench.MATERIAL, case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
ctl = dict(bench.CTL, gnl="GNLY", maxImp="0.0", buckling=False)
st = {}
drv = RiksDriver(m, ctl, rtol=bench.RTOL, stats=st)
drv.start_step()
c = m.ctx
orig_gsm = m.calc_gsm
t_asm = [0.0]


def timed_gsm(*args, **kw):
    c.sync()
    t0 = time.perf_counter()
    out = orig_gsm(*args, **kw)
    c.sync()
    t_asm[0] += time.perf_counter() - t0
    return out


m.calc_gsm = timed_gsm
for k in range(a.steps):
    s0 = dict(st)
    t_asm[0] = 0.0
    c.sync()
    t0 = time.perf_counter()
    drv.iteration()
    c.sync()
    dt = time.perf_counter() - t0
    print("iteration %d: %.1f ms total, assembly + hierarchy refresh %.1f ms, solves %d, PCG iterations %d (in lockstep: %d), solve %.1f ms" %
          (k, dt * 1e3, t_asm[0] * 1e3, st["solves"] - s0["solves"], st["pcg_iterations"] - s0["pcg_iterations"],
           st.get("paired_iterations", 0) - s0.get("paired_iterations", 0), (st["solve_s"] - s0["solve_s"]) * 1e3))
