"""cProfile of the host-sequenced PCG of the multi-GPU layer on ONE rank (1 M plate): where the sequencing thread's time goes per iteration."""
import os, sys, cProfile, pstats, io
import torch, torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.dist import partition_model
from fcvm_hip.loads import Loads
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ["FCVM_DIST_FORCE_HOST"] = "1"
dist.init_process_group("gloo", rank=0, world_size=1)
case = examples.plate_with_hole(float(os.environ.get("PLATE_H", "1.45")))
torch.cuda.set_device(0)
model = partition_model(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]),
                        device_index=0, device=torch.device("cuda", 0))
be = model.ctx
glv, modf, V, _ = model.calc_gsm((0.0, 0.0, 0.0), want_x=False)
f = be.array(model.fixdof.get() * glv.get() + modf.get())
u, info = be.solve(f, rtol=1e-10, maxit=2000)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
u, info = be.solve(f, rtol=1e-10, maxit=2000)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("iterations", info["iterations"])
print(s.getvalue()[:5000])
dist.destroy_process_group()
