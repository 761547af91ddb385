import os, sys, time, cProfile, pstats
REPO="/root/repo"
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel, RiksDriver
case = examples.plate_with_hole(1.45)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
m.ctx.sync()
pr = cProfile.Profile(); pr.enable()
t0=time.perf_counter()
d = RiksDriver(m, case["ctl"], rtol=1e-10, stats={})
m.ctx.sync()
print("RiksDriver init %.3f s" % (time.perf_counter()-t0))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
