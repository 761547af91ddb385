#!/usr/bin/env python3
"""Preconditioner scan on whole analyses (round 3): PCG iterations per solve and solve time of the library's own solver for a list
of cycle settings, on the reference's Plate_with_hole / Embankment examples.

  python tools/precond_probe.py --case plate --h 3.0 --variants add,mult1,mult2
A variant is name[:key=value,...]; keys: additive, k0, alpha0, w (weight of the additive Jacobi part), env.X=... (environment
variable read when the context / hierarchy is created).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))

from fcvm_hip import examples  # noqa: E402
from fcvm_hip.device import Context  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel, calc_disp  # noqa: E402

PRESETS = {
    "add": dict(additive=1),
    "mult1": dict(additive=0, k0=1),
    "mult2": dict(additive=0, k0=2),
    "mult3": dict(additive=0, k0=3, alpha0=8.0),
    "lo": dict(additive=2),
}


def parse(v):
    name, _, rest = v.partition(":")
    cfg = dict(PRESETS.get(name, {}))
    for kv in filter(None, rest.split(";")):
        k, _, val = kv.partition("=")
        cfg[k] = val
    return name if not rest else v, cfg


def run(case, label, cfg, rtol):
    env = {k[4:]: str(v) for k, v in cfg.items() if k.startswith("env.")}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = Context(0)
        ctx.set_precond("mg", k0=int(cfg.get("k0", 1)), alpha0=float(cfg.get("alpha0", 4.0)))
        ctx.set_mg_cycle(int(cfg.get("additive", 1)), float(cfg.get("w", 1.0)))
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                     Loads(ufaces=case["ufaces"], uloads=case["uloads"]) if "ufaces" in case else None, ctx=ctx)
        st = {}
        t0 = time.perf_counter()
        res = calc_disp(m, case["ctl"], rtol=rtol, stats=st)
        ctx.sync()
        dt = time.perf_counter() - t0
        out = dict(variant=label, ne=m.ne, solves=st["solves"], its_per_solve=st["pcg_iterations"] / max(st["solves"], 1),
                   newton=st["newton_iterations"], analysis_s=dt, solve_s=st["solve_s"],
                   ms_per_pcg_iteration=1e3 * st["solve_s"] / max(st["pcg_iterations"], 1), lbd=float(res["lout"][-1]),
                   fine_spmv_total=st.get("spmv_total", 0), precond=ctx.precond_info().get("level2", ""))
        ctx.close()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="plate")
    ap.add_argument("--h", type=float, default=3.0)
    ap.add_argument("--refine", type=int, default=2)
    ap.add_argument("--variants", default="add,mult1,mult2")
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    case = examples.plate_with_hole(a.h) if a.case == "plate" else examples.embankment(a.refine)
    res = []
    for v in a.variants.split(","):
        label, cfg = parse(v)
        try:
            r = run(case, label, cfg, a.rtol)
        except Exception as ex:   # a setting that breaks PCG (not SPD) is a result too
            r = dict(variant=label, error=str(ex)[:300])
        print(json.dumps(r), flush=True)
        res.append(r)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
