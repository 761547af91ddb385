"""what the host-sequenced cycle of the multi-GPU layer costs without any communication: ONE rank on the 1 M plate, the library's own
solver against the same cycle driven from Python (FCVM_DIST_FORCE_HOST=1)"""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.dist import partition_model
from fcvm_hip.loads import Loads
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group(os.environ.get("FCVM_DIST_BACKEND", "gloo"), rank=0, world_size=1)      # (nccl = RCCL: the collectives of the layer on one rank)
if dist.get_backend() == "nccl":
    torch.cuda.set_device(0)
case = examples.plate_with_hole(float(os.environ.get("PLATE_H", "1.45")))
torch.cuda.set_device(0)
for force in os.environ.get("FORCE_MODES", "0,1").split(","):
    os.environ["FCVM_DIST_FORCE_HOST"] = force
    model = partition_model(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]),
                            device_index=0, device=torch.device("cuda", 0))
    be = model.ctx
    glv, modf, V, _ = model.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    f = be.array(model.fixdof.get() * glv.get() + modf.get())
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u, info = be.solve(f, rtol=1e-10, maxit=2000)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("host-sequenced %s: %d iterations, %.1f ms, %.3f ms per iteration %s" % (force, info["iterations"], dt * 1e3, dt * 1e3 / max(info["iterations"], 1),
              info.get("comm_per_iteration", "")), flush=True)
    del model, be
dist.destroy_process_group()
