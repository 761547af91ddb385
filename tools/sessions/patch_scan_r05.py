"""Vertex-patch smoother against block-Jacobi on one elastic solve of a mesh (rtol 1e-10), for a list of environment settings:
PCG iterations, ms per iteration.  usage: patch_scan_r05.py plate:1.45|emb:3 "FCVM_MG_PATCH=1 FCVM_MG_K1S=4" "..." ..."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.device import Context
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
kind, arg = sys.argv[1].split(":")
case = examples.plate_with_hole(float(arg)) if kind == "plate" else examples.embankment(refine=int(arg))
ld = Loads(ufaces=case["ufaces"], uloads=case["uloads"]) if kind == "plate" else Loads()
grav = (0.0, 0.0, 0.0) if kind == "plate" else (0.0, 0.0, -10.0)
for setting in sys.argv[2:]:
    env = dict(kv.split("=") for kv in setting.split())
    for k, v in env.items():
        os.environ[k] = v
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], ld, ctx=Context(0))
    c = m.ctx
    if "SCAN_K1" in env:      # level-1-only cycle with a long Chebyshev sweep as the coarse solve (needs FCVM_MG_LEVEL2=0)
        c.set_precond("mg", k0=1, k1=int(env["SCAN_K1"]), alpha0=4.0, alpha1=float(env.get("SCAN_ALPHA1", "1000")))
    glv, modf, _, _ = m.calc_gsm(grav, want_x=False)
    f = c.axpby(1.0, c.mul(m.fixdof, glv), 1.0, modf)
    u, info = c.solve(f, rtol=1e-10)      # (warm-up: chunk sizes follow the previous solve)
    c.sync(); t0 = time.perf_counter()
    u, info = c.solve(f, rtol=1e-10)
    c.sync(); dt = time.perf_counter() - t0
    pi = c.precond_info()
    print("%s %s [%s]: %d iterations, %.1f ms, %.3f ms per iteration  (patch %s, lmax %.3f, refresh %.1f ms; %d aggregates)" % (
        kind, arg, setting, info["iterations"], dt * 1e3, dt * 1e3 / info["iterations"], pi["patch_smoother"], c.patch_info()["lmax"],
        c.patch_info()["refresh_ms"], pi["l2_aggregates"]), flush=True)
    c.close()
    for k in env:
        del os.environ[k]
