import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fcvm-workbench_amd")
from fcvm_hip import examples
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
case = examples.plate_with_hole(1.45)
from fcvm_hip.device import Context
c0 = Context(0); c0.sync()      # first HIP call of the process outside the measurement, as in bench.py (cube run / membench before?)
for rep in range(2):
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    m.ctx.sync()
    print("GpuModel %.3f s" % (time.perf_counter() - t0))
    pr.disable()
    if rep == 1: pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
    m.ctx.close()
