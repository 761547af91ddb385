#!/usr/bin/env python3
"""Where does the host time of a Newton iteration go?  cProfile of the whole collapse analysis of the 1 M plate (GPU needed).
  python3 tools/newton_host_profile.py [--h 1.45]"""
import argparse, cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel, calc_disp
ap = argparse.ArgumentParser(); ap.add_argument("--h", type=float, default=1.45); a = ap.parse_args()
case = examples.plate_with_hole(a.h)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
st = {}
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
res = calc_disp(m, case["ctl"], rtol=1e-10, stats=st)
m.ctx.sync()
dt = time.perf_counter() - t0
pr.disable()
print("analysis %.3f s, %d Newton iterations, %d solves, %d PCG iterations, solve_s %.3f" % (dt, st["newton_iterations"], st["solves"], st["pcg_iterations"], st["solve_s"]))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
