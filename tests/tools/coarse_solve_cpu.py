#!/usr/bin/env python3
"""CPU prototype (scipy, oracle-assembled K; test infrastructure: uses oracle/): how tight is the COARSE solve of the library's cycle, and what would a
tighter one bring?  The tet4 level is solved by M = Chebyshev(k) block-Jacobi + aggregation with NM modes per aggregate (6 rigid-body modes as in the
library, or the 12 linear fields) + Chebyshev(k); reported: the extreme eigenvalues of M A1 and the PCG iterations of the whole cycle with the
block-Jacobi and the vertex-patch smoother on the fine level.  Round-5 result (plate h = 6, aggregates of ~47 nodes; profiles/r05_patch_smoother.txt):
  exact coarse solve                      eig(M A1) = 1            block-Jacobi 37   vertex patches 26
  Chebyshev(3, 12) + 6 rigid-body modes   [0.11, 1.00]             43                42          <- the library
  the same cycle applied twice            [0.21, 1.00]             39                31
  smoothed aggregation (4/3 / lmax)       [0.20, 1.00]             40                31
  12 linear modes, aggregates of 36       [0.16, 1.00]             38                33
The smallest eigenvalue of M A1 stays at 0.1-0.2 with every two-level aggregation variant: that, not the fine smoother, bounds the cycle.
usage: coarse_solve_cpu.py plate|emb <h | refine> <nodes per aggregate> <6 | 12>"""
import sys, os, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples, meshgen
from oracle import oracle as orc
def pcg(A, b, Bop, rtol=1e-10, maxit=500):
    x = np.zeros_like(b); r = b.copy(); z = Bop(r); p = z.copy(); rz = r @ z; n0 = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q
        if np.linalg.norm(r) <= rtol * n0: return it
        z = Bop(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit
def main(case, arg, target, NM):
    pc = examples.plate_with_hole(float(arg)) if case == "plate" else examples.embankment(refine=int(arg))
    el, xyz = np.asarray(pc["elNodes"]), np.asarray(pc["nocoord"])
    nn = len(xyz); n = 3 * nn
    loads = orc.Loads(ufaces=pc["ufaces"], uloads=pc["uloads"]) if case == "plate" else orc.Loads()
    grav = (0.0, 0.0, 0.0) if case == "plate" else (0.0, 0.0, -10.0)
    stm, row, col, glv, modf, *_ = orc.calcGSM(el, xyz, pc["material"], pc["fix"], grav, loads)
    low = sp.coo_matrix((stm, (row, col)), shape=(n, n)).tocsr()
    K = (low + sp.tril(low, -1).T).tocsr()
    fixed = np.zeros(n, bool); fixed[list(pc["fix"].keys())] = True
    f = np.where(fixed, 0.0, glv) + modf
    e0 = el - 1
    corners = np.unique(e0[:, :4]); f2c = -np.ones(nn, np.int64); f2c[corners] = np.arange(len(corners)); nc = len(corners)
    rows, cols, vals = list(corners), list(f2c[corners]), [1.0] * nc
    pa, pb = -np.ones(nn, np.int64), -np.ones(nn, np.int64)
    for k, (a, b) in enumerate(meshgen.MID_PAIRS):
        pa[e0[:, 4 + k]] = e0[:, a]; pb[e0[:, 4 + k]] = e0[:, b]
    mids = np.where(pa >= 0)[0]
    for m in mids:
        rows += [m, m]; cols += [f2c[pa[m]], f2c[pb[m]]]; vals += [0.5, 0.5]
    P = sp.kron(sp.coo_matrix((vals, (rows, cols)), shape=(nn, nc)).tocsr(), sp.identity(3)).tocsr()
    P = sp.diags((~fixed).astype(float)) @ P
    # level 1 keeps all 3 nc dofs; supported ones get a unit diagonal (as the library's tet4 matrix)
    cfix = np.zeros(3 * nc, bool)
    for c in corners:
        for k in range(3):
            if fixed[3 * c + k]: cfix[3 * f2c[c] + k] = True
    Pf = P @ sp.diags((~cfix).astype(float))
    A1 = (Pf.T @ K @ Pf + sp.diags(cfix.astype(float))).tocsr()
    n1 = 3 * nc
    Mex = spla.factorized(A1.tocsc())
    D1inv = sp.block_diag([np.linalg.inv(A1[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray()) for i in range(nc)]).tocsr()
    l1 = 1.02 * spla.eigsh(spla.LinearOperator((n1, n1), matvec=lambda v: D1inv @ (A1 @ v)), k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0]
    # aggregates: uniform bins over the bounding box
    x1 = xyz[corners]
    lo, hi = x1.min(0), x1.max(0)
    nb_target = max(1, nc // target)
    ext = np.maximum(hi - lo, 1e-9); cell = (np.prod(ext) / nb_target) ** (1 / 3)
    nbin = np.maximum(1, np.round(ext / cell).astype(int))
    ib = np.minimum(((x1 - lo) / ext * nbin).astype(int), nbin - 1)
    key = (ib[:, 0] * nbin[1] + ib[:, 1]) * nbin[2] + ib[:, 2]
    _, agg = np.unique(key, return_inverse=True); nagg = agg.max() + 1
    cen = np.zeros((nagg, 3)); np.add.at(cen, agg, x1); cen /= np.bincount(agg)[:, None]
    R, C, V = [], [], []
    hs = cell
    for i in range(nc):
        d = x1[i] - cen[agg[i]]
        if NM == 6:
            blk = np.zeros((3, 6)); blk[:, :3] = np.eye(3)
            blk[:, 3:] = np.array([[0, d[2], -d[1]], [-d[2], 0, d[0]], [d[1], -d[0], 0]])
        else:
            blk = np.zeros((3, 12)); blk[:, :3] = np.eye(3)
            for jj in range(3):
                blk[:, 3 + 3 * jj:6 + 3 * jj] = d[jj] / hs * np.eye(3)
        for r in range(3):
            for c in range(NM):
                if blk[r, c] != 0 and not cfix[3 * i + r]:
                    R.append(3 * i + r); C.append(NM * agg[i] + c); V.append(blk[r, c])
    Z = sp.coo_matrix((V, (R, C)), shape=(n1, NM * nagg)).tocsr()
    print(f"{case} {arg}: {len(el)} tet10, {nc} corners, {nagg} aggregates (~{nc / nagg:.0f} nodes), level-1 lmax {l1:.3f}", flush=True)
    def coarse_solver(Zm):
        A2 = (Zm.T @ A1 @ Zm).toarray()
        w, U = np.linalg.eigh(A2)
        keep = w > 1e-10 * w.max()
        A2i = (U[:, keep] / w[keep]) @ U[:, keep].T
        return lambda r2: A2i @ r2
    def cheb(k, alpha):
        lmin = l1 / alpha; theta = 0.5 * (l1 + lmin); delta = 0.5 * (l1 - lmin); sigma = theta / delta
        def run(x, r):       # k steps on A1 e = r starting from x (updates x, r)
            d = (D1inv @ r) / theta; rho = 1.0 / sigma
            for it in range(k):
                x = x + d; r = r - A1 @ d
                if it + 1 < k:
                    rn = 1.0 / (2.0 * sigma - rho)
                    d = rn * rho * d + 2.0 * rn / delta * (D1inv @ r); rho = rn
            return x, r
        return run
    def level1_cycle(Zm, k, alpha, times=1):
        A2s = coarse_solver(Zm); sm = cheb(k, alpha)
        def once(b):
            x = np.zeros_like(b); r = b.copy()
            if k: x, r = sm(x, r)
            e = Zm @ A2s(Zm.T @ r); x = x + e; r = r - A1 @ e
            if k: x, r = sm(x, r)
            return x
        def M(b):
            x = once(b)
            for _ in range(times - 1):
                x = x + once(b - A1 @ x)
            return x
        return M
    # fine smoothers
    def lam_max(Sop): return spla.eigsh(spla.LinearOperator((n, n), matvec=lambda v: Sop(K @ v)), k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0]
    def weight(lm): return 2.0 / (1.02 * lm * (1 + 1 / 16.0))
    Dinv = sp.block_diag([np.linalg.inv(K[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray()) for i in range(nn)]).tocsr()
    SJ = lambda r: Dinv @ r; wJ = weight(lam_max(SJ))
    star = [[] for _ in range(nn)]
    for m in mids: star[pa[m]].append(m); star[pb[m]].append(m)
    Rr, Cc, Vv = [], [], []
    for c in corners:
        idx = (3 * np.asarray([c] + star[c])[:, None] + np.arange(3)[None, :]).ravel()
        Ki = np.linalg.inv(K[idx][:, idx].toarray())
        rr, cc = np.meshgrid(idx, idx, indexing="ij"); Rr.append(rr.ravel()); Cc.append(cc.ravel()); Vv.append(Ki.ravel())
    Sp = sp.coo_matrix((np.concatenate(Vv), (np.concatenate(Rr), np.concatenate(Cc))), shape=(n, n)).tocsr()
    SV = lambda r: Sp @ r; wV = weight(lam_max(SV))
    def outer(Sop, w, Mc):
        def B(r):
            z0 = w * Sop(r); r1 = r - K @ z0; z = z0 + Pf @ Mc(Pf.T @ r1)
            return z + w * Sop(r - K @ z)
        return B
    def report(name, Mc):
        # spectrum of Mc A1 on random vectors: extreme Ritz values by Lanczos (eigsh on the symmetrised operator is awkward: use power-type estimates)
        op = spla.LinearOperator((n1, n1), matvec=lambda v: Mc(A1 @ v))
        try:
            lmx = spla.eigs(op, k=1, which="LR", return_eigenvectors=False, tol=1e-2, maxiter=300)[0].real
            lmn = spla.eigs(op, k=1, which="SR", return_eigenvectors=False, tol=1e-2, maxiter=2000)[0].real
        except Exception as e:
            lmx = lmn = float("nan")
        print(f"  {name}: eig(M A1) in [{lmn:.3f}, {lmx:.3f}]   outer PCG: block-Jacobi {pcg(K, f, outer(SJ, wJ, Mc))}, vertex patches {pcg(K, f, outer(SV, wV, Mc))}", flush=True)
    print(f"  modes per aggregate {NM}, unknowns {NM * nagg}")
    report("exact coarse solve", Mex)
    report("Chebyshev(3, 12) + aggregation", level1_cycle(Z, 3, 12.0))
    report("Chebyshev(3, 12) + aggregation, applied twice", level1_cycle(Z, 3, 12.0, times=2))
    report("Chebyshev(4, 16) + aggregation", level1_cycle(Z, 4, 16.0))
    Zs = (Z - (4.0 / 3.0 / (l1 / 1.02)) * (D1inv @ (A1 @ Z))).tocsr()
    report(f"smoothed aggregation (4/3 / lmax, nnz(Zs) / nnz(Z) = {Zs.nnz / Z.nnz:.1f}) + Chebyshev(3, 12)", level1_cycle(Zs, 3, 12.0))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
