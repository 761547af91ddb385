#!/usr/bin/env python3
"""Whole elastoplastic collapse analysis of one of the reference's examples on one GPU; prints the load
history, timing and solver statistics.
   python tools/example_run.py plate --h 2.9          # Plate_with_hole_Example, ~50k tet10 (BASELINE configs[1])
   python tools/example_run.py plate --h 1.1          # ~1M tet10 (north-star size)
   python tools/example_run.py embankment --refine 3"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel, calc_disp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("example", choices=["plate", "embankment"])
ap.add_argument("--h", type=float, default=2.9, help="plate: edge length of the mapped mesh")
ap.add_argument("--refine", type=int, default=2, help="embankment: uniform refinements")
ap.add_argument("--nstep", type=int, default=0)
ap.add_argument("--gnl", type=str, default="", help="GNLY: geometrically non-linear (tangent re-assembly every Newton iteration)")
a = ap.parse_args()
t0 = time.perf_counter()
if a.example == "plate":
    case = examples.plate_with_hole(a.h)
    loads = Loads(ufaces=case["ufaces"], uloads=case["uloads"])
else:
    case = examples.embankment(a.refine)
    loads = Loads()
if a.nstep:
    case["ctl"]["nstep"] = a.nstep
if a.gnl:
    case["ctl"]["gnl"] = a.gnl
    case["ctl"]["maxImp"] = "0.0"
t_mesh = time.perf_counter() - t0
t0 = time.perf_counter()
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], loads)
t1 = time.perf_counter()
st = {}
last = [time.perf_counter()]


def log(msg):
    if msg.startswith("Step"):
        print(msg, "  (%.1fs)" % (time.perf_counter() - last[0]), flush=True)


res = calc_disp(m, case["ctl"], log=log, stats=st)
t2 = time.perf_counter()
print("load factors", np.round(res["lout"], 4).tolist())
print("max |u|     ", np.round(res["un"], 4).tolist())
print("%s: %d tet10, %d dof; mesh %.2fs; set-up %.2fs; analysis %.2fs; steps %d; Newton iterations %d (%.1f /s); solves %d; "
      "PCG iterations %d (%.1f per solve); solve time %.2fs (fine SpMV %.3f ms each); load factor %.4f; max |u| %.3f; %s" % (
          case["name"], m.ne, m.ndof, t_mesh, t1 - t0, t2 - t1, len(res["lout"]) - 1, st["newton_iterations"],
          st["newton_iterations"] / (t2 - t1), st["solves"], st["pcg_iterations"], st["pcg_iterations"] / max(st["solves"], 1),
          st["solve_s"], st.get("spmv_ms", 0.0) / max(st.get("spmv_launches", 0), 1), res["lout"][-1], res["un"][-1], m.ctx.precond_info()))
