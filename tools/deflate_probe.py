#!/usr/bin/env python3
"""Solver-space experiment (round 3): what does recycling across the constant-matrix solves of a GNLN analysis buy?

A Python-driven PCG (library SpMV + library V-cycle through fcvm_mg_apply, host scalars) that can
  * record the Lanczos data of a solve (alpha, beta, the M-orthonormal z_j),
  * form the k lowest Ritz vectors W of M^-1 K from it, K W and E = W^T K W,
  * start later solves from the Galerkin projection onto W ("init") and / or keep every search direction K-orthogonal to W
    ("deflate": Saad, Yeung, Erhel, Guyomarc'h, SIAM J. Sci. Comput. 21 (2000)).
Iteration counts are what this tool is for; its timings are not (every dot is a host synchronisation).

  python tools/deflate_probe.py --case plate --h 3.0 --variants base,init16,defl16
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
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel, RiksDriver  # noqa: E402


class Space:
    """deflation space: W, K W, Cholesky factor of E = W^T K W"""

    def __init__(self, c, W):
        self.c, self.W = c, W
        self.KW = [c.spmv(w, c.empty(w.n)) for w in W]
        k = len(W)
        E = np.zeros((k, k))
        for i in range(k):
            E[i, :] = c.mdot(W, self.KW[i])
        self.E = 0.5 * (E + E.T)
        self.L = np.linalg.cholesky(self.E)

    def esolve(self, b):
        y = np.linalg.solve(self.L, np.asarray(b))
        return np.linalg.solve(self.L.T, y)

    def combine(self, coef, out=None, sign=1.0):
        c = self.c
        for a, w in zip(coef, self.W):
            if out is None:
                out = c.axpby(sign * a, w, 0.0, None)
            else:
                c.axpby(1.0, out, sign * a, w, out=out)
        return out


def pcg(c, b, rtol, x0=None, space=None, deflate=False, record=False, maxit=2000):
    """returns (x, iterations, lanczos) ; lanczos = (alphas, betas, [z_j / sqrt(r_j.z_j)]) when record"""
    n = b.n
    x = x0.clone() if x0 is not None else c.zeros(n)
    if x0 is not None:
        r = c.spmv(x, c.empty(n))
        c.axpby(1.0, b, -1.0, r, out=r)
    else:
        r = b.clone()
    bb = c.dot(b, b)
    if space is not None:   # x += W E^-1 W^T r
        a = space.esolve(c.mdot(space.W, r))
        space.combine(a, out=x)
        r = c.spmv(x, c.empty(n))
        c.axpby(1.0, b, -1.0, r, out=r)
    z, q = c.empty(n), c.empty(n)
    c.mg_apply(r, z)
    rz, rr = c.mdot([z, r], r)
    p = z.clone()
    if space is not None and deflate:
        mu = space.esolve(c.mdot(space.KW, z))
        space.combine(mu, out=p, sign=-1.0)
    al, be, Z = [], [], []
    it = 0
    hist = [np.sqrt(rr / bb)]
    while rr > rtol * rtol * bb and it < maxit:
        if record:
            Z.append(c.axpby(1.0 / np.sqrt(rz), z, 0.0, None))
        c.spmv(p, q)
        pq = c.dot(p, q)
        alpha = rz / pq
        c.axpby(1.0, x, alpha, p, out=x)
        c.axpby(1.0, r, -alpha, q, out=r)
        c.mg_apply(r, z)
        rz_new, rr = c.mdot([z, r], r)
        beta = rz_new / rz
        rz = rz_new
        c.axpby(1.0, z, beta, p, out=p)
        if space is not None and deflate:
            mu = space.esolve(c.mdot(space.KW, z))
            space.combine(mu, out=p, sign=-1.0)
        al.append(alpha); be.append(beta)
        it += 1
        hist.append(np.sqrt(rr / bb))
    return x, it, (al, be, Z, hist)


def lanczos_T(al, be):
    m = len(al)
    T = np.zeros((m, m))
    for j in range(m):
        T[j, j] = 1.0 / al[j] + (be[j - 1] / al[j - 1] if j > 0 else 0.0)
        if j + 1 < m:
            T[j, j + 1] = T[j + 1, j] = -np.sqrt(be[j]) / al[j]
    return T


def ritz_space(c, lan, k):
    al, be, Z, _ = lan
    T = lanczos_T(al, be)
    th, S = np.linalg.eigh(T)
    W = []
    for i in range(min(k, len(th))):
        w = None
        for j, zj in enumerate(Z):
            w = c.axpby(S[j, i], zj, 0.0, None) if w is None else c.axpby(1.0, w, S[j, i], zj, out=w)
        W.append(w)
    return th, W


class ProbeDriver(RiksDriver):
    def __init__(self, model, ctl, variant, rtol=1e-10):
        self.variant = variant
        self.k = int("".join(ch for ch in variant if ch.isdigit()) or 0)
        self.mode = "".join(ch for ch in variant if not ch.isdigit())
        self.space = None
        self.iters = []
        self.theta = None
        super().__init__(model, ctl, rtol=rtol)

    def solve(self, rhs, guess=None):
        if self.mode == "base":
            n0 = self.st["pcg_iterations"]
            u = super().solve(rhs, guess=guess)
            self.iters.append(self.st["pcg_iterations"] - n0)
            return u
        c = self.c
        first = self.space is None and self.mode in ("init", "defl", "initproj")
        x, it, lan = pcg(c, rhs, self.rtol, space=self.space, deflate=self.mode == "defl", record=first)
        if first:
            self.theta, W = ritz_space(c, lan, self.k)
            self.space = Space(c, W)
            del lan
        self.iters.append(it)
        self.st["pcg_iterations"] += it
        self.st["solves"] += 1
        return x


def run(case, variant, rtol):
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                 Loads(ufaces=case.get("ufaces"), uloads=case.get("uloads")) if "ufaces" in case else None)
    d = ProbeDriver(m, case["ctl"], variant, rtol=rtol)
    t0 = time.perf_counter()
    cnt = not d.elastic_only
    pstep = 0
    while cnt and pstep < d.nstep and not d.mrr:
        pstep += 1
        error = d.start_step()
        while error > d.error_max and not d.mrr:
            error = d.iteration()
            if d.iterat > d.iterat_max:
                d.restart()
                error = d.error
        if d.end_step():
            break
    out = dict(variant=variant, ne=m.ne, solves=len(d.iters), iters=d.iters, mean=float(np.mean(d.iters)),
               lbd=float(d.lbd[-1]), newton=d.st["newton_iterations"], wall_s=time.perf_counter() - t0)
    if d.theta is not None:
        out["ritz_lowest"] = [float(v) for v in d.theta[:40]]
        out["ritz_highest"] = [float(v) for v in d.theta[-5:]]
    m.ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="plate")
    ap.add_argument("--h", type=float, default=3.0)
    ap.add_argument("--refine", type=int, default=2)
    ap.add_argument("--variants", default="base,init16,defl16")
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    case = examples.plate_with_hole(a.h) if a.case == "plate" else examples.embankment(a.refine)
    res = []
    for v in a.variants.split(","):
        r = run(case, v, a.rtol)
        print(json.dumps(r), flush=True)
        res.append(r)
    if a.out:
        json.dump(res, open(a.out, "w"))


if __name__ == "__main__":
    main()
