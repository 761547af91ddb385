#!/usr/bin/env python3
"""Preconditioner probe: solves K u = f on the synthetic cube with block-Jacobi and with the
p-multigrid V-cycle for several smoother settings; prints iterations, time and SpMV counts."""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import meshgen  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", choices=["cube", "plate", "box", "brick"], default="cube")
ap.add_argument("--dims", type=str, default="20,20,160", help="brick: cells in x,y,z (cell size 100/20), clamped at z = 0")
ap.add_argument("--h", type=float, default=2.9, help="plate: mesh edge length")
ap.add_argument("--cells", type=int, default=24)
ap.add_argument("--rtol", type=float, default=1e-10)
ap.add_argument("--configs", type=str, default="bj;2,30,8,300;3,30,10,300;2,60,8,1000;3,60,10,1000")
ap.add_argument("--nu", type=float, default=0.3)
ap.add_argument("--l2", type=str, default="", help="level-2 settings target,k1s,alpha1s;... appended to each mg config")
a = ap.parse_args()
if a.case == "plate":
    from fcvm_hip import examples
    case = examples.plate_with_hole(a.h)
elif a.case == "box":  # the plate without the hole (undistorted Kuhn cells), same supports and load
    nx, ny, nz = int(round(200 / a.h)), int(round(100 / a.h)), max(2, int(round(10 / a.h)))
    el, xyz = meshgen.cube_tet10(nx, ny, nz, 200.0, 100.0, 10.0)
    fix = {3 * int(n): 0.0 for n in meshgen.nodes_on_plane(xyz, 0, 0.0)}
    fix.update({3 * int(n) + 1: 0.0 for n in meshgen.nodes_on_plane(xyz, 0, 200.0)})
    for pt in ((0.0, 0.0, 0.0), (0.0, 100.0, 0.0), (200.0, 0.0, 0.0)):
        fix[3 * int(np.argmin(np.abs(xyz - np.array(pt)).max(1))) + 2] = 0.0
    faces = meshgen.boundary_faces_on_plane(el, xyz, 0, 200.0)
    case = dict(elNodes=el, nocoord=xyz, fix=fix, ufaces=faces, uloads=np.tile([100.0, 0.0, 0.0], (len(faces), 1)))
elif a.case == "brick":  # weak-scaling layouts of the distributed bench, solved on one GPU
    nx, ny, nz = [int(v) for v in a.dims.split(",")]
    el, xyz = meshgen.cube_tet10(nx, ny, nz, 5.0 * nx, 5.0 * ny, 5.0 * nz)
    fix = {3 * int(n) + k: 0.0 for n in meshgen.nodes_on_plane(xyz, 2, 0.0) for k in range(3)}
    faces = meshgen.boundary_faces_on_plane(el, xyz, 2, 5.0 * nz)
    case = dict(elNodes=el, nocoord=xyz, fix=fix, ufaces=faces, uloads=np.tile([20.0 * nx / nz, 0.0, 200.0], (len(faces), 1)))
else:
    case = meshgen.cube_case(a.cells, l=100.0, traction=(20.0, 0.0, 200.0))
m = GpuModel(case["elNodes"], case["nocoord"], (210000.0, a.nu, 7.8e-6), case["fix"],
             Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
print("elements", m.ne, "dof", m.ndof, flush=True)
ref = None
os.environ["FCVM_MG_LEVEL2"] = "0"
runs = []
for cfg in a.configs.split(";"):
    if cfg == "bj" or not a.l2:
        runs.append((cfg, None))
    for l2 in [x for x in a.l2.split(";") if x]:
        if cfg != "bj":
            runs.append((cfg, l2))
for cfg, l2 in runs:
    if cfg == "bj":
        c.set_precond("bj")
    else:
        k0, k1, a0, a1 = cfg.split(",")[:4]
        cyc = cfg.split(",")[4:]  # optional: a|m (additive / multiplicative), weight
        c.set_mg_cycle(additive=(not cyc or cyc[0] == "a"), weight=float(cyc[1]) if len(cyc) > 1 else 1.0)
        c.set_precond("mg", int(k0), int(k1), float(a0), float(a1))
    t0 = time.perf_counter()
    glv, modf, V, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    c.sync()
    t_setup = time.perf_counter() - t0
    if l2:
        t0 = time.perf_counter()
        if l2 == "auto":  # the product's default (device.Context.build_level2)
            li = c.build_level2()
        else:
            tg, k1s, a1s = l2.split(",")
            li = c.build_level2(target=int(tg), k1s=int(k1s), alpha1s=float(a1s))
        cfg = cfg + "|L2 " + l2 + " " + str(li) + " %.1fs" % (time.perf_counter() - t0)
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    c.sync()
    t0 = time.perf_counter()
    try:
        u, info = c.solve(f, rtol=a.rtol, maxit=20000)
    except Exception as e:
        print(cfg, "FAILED", str(e)[:120], flush=True)
        continue
    c.sync()
    dt = time.perf_counter() - t0
    tm = c.solve_timing()
    uh = u.get()
    if ref is None:
        ref = uh
    # true residual with an independent SpMV
    r = c.axpby(1.0, f, -1.0, c.spmv(u, c.empty(m.ndof)))
    print("%-16s setup %.2fs  iters %5d  solve %.3fs  fine spmv %6d  coarse spmv %6d  spmv avg %.3f ms  true relres %.2e  |u-u_bj|/|u| %.2e  %s" % (
        cfg, t_setup, info["iterations"], dt, info["spmv"], tm["coarse_spmv"], tm["spmv_ms"] / max(tm["spmv_timed"], 1),
        c.norm(r) / c.norm(f), np.abs(uh - ref).max() / np.abs(ref).max(), c.precond_info() if cfg != "bj" else ""), flush=True)
