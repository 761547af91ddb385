#!/usr/bin/env python3
"""Debug rehearsal of the distributed driver: 2 ranks sharing GPU 0 over gloo, progress to files."""
import faulthandler
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def worker(rank, world, port):
    out = open(os.path.join(REPO, "gpurun_out", "rehearsal_rank%d.log" % rank), "w", buffering=1)
    faulthandler.enable(out)
    faulthandler.dump_traceback_later(45, repeat=True, file=out)
    def say(*a):
        print("[%.1f]" % time.time(), *a, file=out, flush=True)
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    say("init pg")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    say("pg ok")
    torch.cuda.set_device(0)
    from fcvm_hip.dist_bench import build_rank_model
    from fcvm_hip.solver import RiksDriver
    import test_gpu_dist as T
    model, be, ops = build_rank_model(rank, world, T.CELLS, T.MAT, T.TRACTION, 0, torch.device("cuda", 0), length=10.0)
    say("model built nn", model.nn, "shared", {k: len(v) for k, v in model.part.shared.items()})
    st = {}
    drv = RiksDriver(model, T.CTL, rtol=1e-12, maxit=4000, stats=st, log=say)
    say("driver ready", st)
    e = drv.start_step()
    say("start_step error", e)
    for i in range(3):
        e = drv.iteration()
        say("iteration", i, e, st)
    dist.destroy_process_group()
    say("done")


if __name__ == "__main__":
    import torch.multiprocessing as mp
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    mp.spawn(worker, args=(2, 29544), nprocs=2, join=True)
