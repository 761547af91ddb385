#!/usr/bin/env python3
"""Where the one-time set-up of an analysis goes (pattern, assembly, multigrid hierarchy, aggregate level):
   python tools/setup_profile.py [--cells 55 | --plate-h 1.45]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples, meshgen  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=55)
ap.add_argument("--plate-h", type=float, default=0.0)
a = ap.parse_args()
t0 = time.perf_counter()
if a.plate_h:
    case = examples.plate_with_hole(a.plate_h)
    mat = case["material"]
else:
    case = meshgen.cube_case(a.cells, l=100.0, traction=(20.0, 0.0, 200.0))
    mat = (210000.0, 0.3, 7.8e-6)
t1 = time.perf_counter()
m = GpuModel(case["elNodes"], case["nocoord"], mat, case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
m.ctx.sync()
t2 = time.perf_counter()
print("mesh %.2fs  GpuModel (pattern + upload) %.2fs (pattern %.2fs)" % (t1 - t0, t2 - t1, m.t_pattern), flush=True)
os.environ["FCVM_MG_LEVEL2"] = "0"
t0 = time.perf_counter()
m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
m.ctx.sync()
t1 = time.perf_counter()
print("first calc_gsm without the aggregate level (assembly + tet4 level + eigenvalue estimates) %.2fs" % (t1 - t0), flush=True)
pr = cProfile.Profile()
pr.enable()
info = m.ctx.build_level2()
m.ctx.sync()
pr.disable()
print("build_level2", info, "%.2fs" % (time.perf_counter() - t1), flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
