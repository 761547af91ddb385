"""does a 12 M tet10 cube still fit / run after round 4's additions (assembly tables of the symmetric form)?  python big_cube_r04.py [cells]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
import bench
from fcvm_hip import meshgen
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 126
t0 = time.perf_counter()
case = meshgen.cube_case(cells, l=100.0, traction=bench.TRACTION)
print("mesh %d tet10, %d nodes: %.1f s" % (len(case["elNodes"]), len(case["nocoord"]), time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
m = GpuModel(case["elNodes"], case["nocoord"], bench.MATERIAL, case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
print("GpuModel %.1f s; free/total GB %s" % (time.perf_counter() - t0, [round(v / 1e9, 1) for v in c.mem_info()]), flush=True)
t0 = time.perf_counter()
glv, modf, V, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
c.sync()
print("calc_gsm (assembly + hierarchy) %.2f s, volume %.6g; %s; free/total GB %s" % (time.perf_counter() - t0, V, c.spmv_sym_info(), [round(v / 1e9, 1) for v in c.mem_info()]), flush=True)
f = c.mul(m.fixdof, glv)
t0 = time.perf_counter()
u, info = c.solve(f, rtol=1e-10)
c.sync()
print("solve: %d iterations, %.2f s; residual check %.3e" % (info["iterations"], time.perf_counter() - t0,
      c.norm(c.axpby(1.0, f, -1.0, c.spmv(u, c.empty(m.ndof)))) / c.norm(f)), flush=True)
for _ in range(2):
    c.sync(); t0 = time.perf_counter(); m.calc_gsm((0.0, 0.0, 0.0), want_x=False); c.sync()
    print("calc_gsm again %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
