#!/usr/bin/env python3
"""Debug aid for the distributed multigrid: N ranks sharing GPU 0 over gloo, one elastic solve with a bounded
iteration count, aggregate-level diagnostics.   python tools/dist_debug.py <ranks> <cells> [target]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))


def worker(rank, world, port, cells, target):
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from fcvm_hip import dist as D
    from fcvm_hip.dist_bench import build_rank_model
    if target:
        orig = D.DistMG.__init__

        def patched(self, *a, **k):
            k["target"] = target
            orig(self, *a, **k)
        D.DistMG.__init__ = patched
    t0 = time.time()
    model, be, ops = build_rank_model(rank, world, cells, (210000.0, 0.3, 7.8e-6), (20.0, 0.0, 200.0), 0, torch.device("cuda", 0))
    glv, modf, V, _ = model.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    f = be.array(model.fixdof.get() * glv.get() + modf.get())
    torch.cuda.synchronize()
    t1 = time.time()
    try:
        u, info = be.solve(f, rtol=1e-10, maxit=200)
        msg = "converged %s" % info
    except Exception as e:
        msg = "FAILED %r" % (e,)
    torch.cuda.synchronize()
    if rank == 0:
        print("ranks %d cells %d: set-up %.1fs solve %.1fs  l2 %s lmax %.3f %.3f  %s" % (
            world, cells, t1 - t0, time.time() - t1, be.mg.l2_info, be.mg.lmax0, be.mg.lmax1, msg), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    import torch.multiprocessing as mp
    world, cells = int(sys.argv[1]), int(sys.argv[2])
    target = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(world, port, cells, target), nprocs=world, join=True)
