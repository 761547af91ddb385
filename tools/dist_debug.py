#!/usr/bin/env python3
"""Where does a partitioned run spend its time?  N ranks on GPU 0 (gloo), the Embankment model refined `--refine` times,
every solve logged per rank (iterations, residual, seconds) to gpurun_out/<tag>/dist_rank<r>.log, tracebacks of all threads
after --dump seconds.

  timeout -k 10 300 python3 tools/dist_debug.py --ranks 3 --refine 1 --nstep 8 --tag r02l
"""
import argparse
import faulthandler
import os
import socket
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))


def worker(rank, world, port, a):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.makedirs(os.path.join(REPO, "gpurun_out", a.tag), exist_ok=True)
    log = open(os.path.join(REPO, "gpurun_out", a.tag, "dist_rank%d.log" % rank), "w", buffering=1)
    faulthandler.dump_traceback_later(a.dump, repeat=True, file=log)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fcvm_hip import examples
        from fcvm_hip.dist import partition_model
        from fcvm_hip.loads import Loads
        from fcvm_hip.solver import calc_disp
        torch.cuda.set_device(0)
        case = examples.embankment(refine=a.refine)
        t0 = time.perf_counter()
        model = partition_model(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(), 0)
        be = model.ctx
        print("rank %d: %d elements, %d nodes, set-up %.2f s" % (rank, model.ne, model.nn, time.perf_counter() - t0), file=log)
        orig = be.solve

        def solve(f, u=None, rtol=1e-10, maxit=20000):
            t = time.perf_counter()
            out = orig(f, u=u, rtol=rtol, maxit=maxit)
            torch.cuda.synchronize()
            print("solve: %s  %.3f s  l2 %s" % (out[1], time.perf_counter() - t, getattr(be.mg, "l2_info", None)), file=log)
            return out
        be.solve = solve
        import cProfile
        import pstats
        pr = cProfile.Profile() if a.profile else None
        if pr:
            pr.enable()
        res = calc_disp(model, dict(case["ctl"], nstep=a.nstep), rtol=1e-11, maxit=a.maxit,
                        log=lambda *m: print(*m, file=log))
        if pr:
            pr.disable()
            pstats.Stats(pr, stream=log).sort_stats("tottime").print_stats(30)
        print("done: lout", res["lout"], "pcg", res["stats"]["pcg_iterations"], "solves", res["stats"]["solves"],
              "%.1f s" % (time.perf_counter() - t0), file=log)
    finally:
        faulthandler.cancel_dump_traceback_later()
        dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=3)
    ap.add_argument("--refine", type=int, default=1)
    ap.add_argument("--nstep", type=int, default=8)
    ap.add_argument("--maxit", type=int, default=1500)
    ap.add_argument("--dump", type=float, default=60.0)
    ap.add_argument("--tag", default="dist_debug")
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(a.ranks, port, a), nprocs=a.ranks, join=True)
    print("dist_debug: finished")
