#!/usr/bin/env python3
"""Sparse aggregate level (fcvm_l2s.inc) forced on a mid-size mesh: is the V-cycle symmetric positive definite, how many PCG
iterations does it take against the dense aggregate level?   python3 tools/l2s_probe.py [--max-aggr 40]"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--max-aggr", type=int, default=40)
ap.add_argument("--refine", type=int, default=2)
ap.add_argument("--rtol", type=float, default=1e-10)
ap.add_argument("--cube", type=int, default=0, help="cells per edge of the synthetic cube instead of the Embankment model")
a = ap.parse_args()
from fcvm_hip import examples  # noqa: E402
from fcvm_hip.device import Context  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

if a.cube:
    from fcvm_hip import meshgen
    case = meshgen.cube_case(a.cube, l=100.0, traction=(20.0, 0.0, 200.0))
    case["material"] = (210000.0, 0.3, 7.8e-6)
    LOADS = Loads(ufaces=case["ufaces"], uloads=case["uloads"])
    GRAV = (0.0, 0.0, 0.0)
else:
    case = examples.embankment(refine=a.refine)
    LOADS, GRAV = Loads(), (0.0, 0.0, -10.0)
os.environ["FCVM_VERBOSE"] = "1"


def run(tag, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], LOADS, ctx=Context(0))
        c = m.ctx
        glv, modf, _, _ = m.calc_gsm(GRAV, want_x=False)
        rng = np.random.default_rng(1)
        x, y = c.array(m.fixdof_h * rng.standard_normal(m.ndof)), c.array(m.fixdof_h * rng.standard_normal(m.ndof))
        mx, my = c.precond(x, c.empty(m.ndof)), c.precond(y, c.empty(m.ndof))
        sym = abs(c.dot(y, mx) - c.dot(x, my)) / abs(c.dot(y, mx))
        print("%-34s %s" % (tag, c.precond_info()["level2"]))
        print("%-34s x.Mx = %.3e  y.My = %.3e  asymmetry %.1e" % ("", c.dot(x, mx), c.dot(y, my), sym))
        f = c.mul(m.fixdof, glv)
        try:
            u, info = c.solve(f, rtol=a.rtol)
            print("%-34s PCG iterations %d" % ("", info["iterations"]))
        except Exception as e:
            print("%-34s solve failed: %s" % ("", str(e)[:100]))
        c.close()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


run("dense aggregate level", {})
M = str(a.max_aggr)
run("dense, 8x larger aggregates", {"FCVM_MG_TARGET": str(8 * 176)})
run("sparse, 8 steps", {"FCVM_MG_MAX_AGGREGATES": M})
run("sparse, 8 steps, 2 coarse corrections", {"FCVM_MG_MAX_AGGREGATES": M, "FCVM_L2S_COARSE": "2"})
run("sparse, 4 steps, 2 coarse corrections", {"FCVM_MG_MAX_AGGREGATES": M, "FCVM_L2S_COARSE": "2", "FCVM_L2S_K": "4", "FCVM_L2S_ALPHA": "30"})
run("sparse, 4 steps, 3 coarse corrections", {"FCVM_MG_MAX_AGGREGATES": M, "FCVM_L2S_COARSE": "3", "FCVM_L2S_K": "4", "FCVM_L2S_ALPHA": "30"})
run("sparse, 16 steps, alpha 300", {"FCVM_MG_MAX_AGGREGATES": M, "FCVM_L2S_K": "16", "FCVM_L2S_ALPHA": "300"})
