import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fcvm-workbench_amd")
from fcvm_hip import examples
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel, calc_disp
from fcvm_hip.device import Context
c = Context(0)
f0, tot = c.mem_info()
case = examples.plate_with_hole(1.45)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctx=c)
f1, _ = c.mem_info()
st = {}
res = calc_disp(m, case["ctl"], rtol=1e-10, stats=st)
c.sync()
f2, _ = c.mem_info()
print("device memory: after set_mesh %.2f GB, after the analysis %.2f GB (of %.0f GB)" % ((f0 - f1) / 1e9, (f0 - f2) / 1e9, tot / 1e9))
print(c.spmv_sym_info(), c.spmv_info()["bytes"] / 1e9)
