#!/usr/bin/env python3
"""CPU prototype (scipy, oracle-assembled K; test infrastructure: uses oracle/): what the FINE-LEVEL SMOOTHER of the two-level tet10 -> tet4
cycle is worth.  PCG iterations (rtol 1e-10) of the multiplicative V(1,1) cycle  x1 = w S b; x2 = x1 + P M P^T (b - K x1); x = x2 + w S (b - K x2)
with
  S = nodal block-Jacobi (the library's smoother), w = degree-1 Chebyshev weight on [lmax / 16, lmax] of S K
  S = additive Schwarz over VERTEX PATCHES (a corner node + the mid-side nodes of its edges; fcvm_patch_kernels.hip.h)
  S = edge patches (a mid-side node + its two corners), edge stars + corner Jacobi, partition-of-unity scaled patches
  patch inverses rounded to FP32 / 16-bit fixed point / FP16 / bf16, smoother matrices K rounded to b bits (why the 16-bit passes cannot go to 8)
with M = exact coarse solve and, for the question that decided the round-5 experiment, M = an inexact one (k Chebyshev steps of point Jacobi).
Round-5 results (profiles/r05_patch_smoother.txt):
  plate h = 6 / 4 (16,830 / 48,600 tet10), exact M: block-Jacobi 38 / 37, vertex patches 25 / 25; Embankment refine 1 / 2: 54 / 65 -> 32 / 40;
  FP32, FP16 (per-patch scale) and bf16 inverses: unchanged; 16-bit fixed point per patch: breaks down (supports next to stiffness entries);
  inexact M (Chebyshev(60, 1e3) / (200, 1e4) on the plate at h = 8): 78 -> 63 / 53 -> 41 -- the gain is ~13 iterations whatever the level,
  and with the library's aggregate-based coarse solve it is 41 -> 34 (7 k tet10), 42 -> 40 (48.6 k), 46 -> 44 (1 M): the coarse solve bounds the cycle.
  smoother matrix K rounded to 16 / 12 / 10 / 8 bits per 3 x 3 block: 38 / 43 / 70 / 151 iterations (plate h = 4).
usage: patch_smoother_cpu.py plate|emb <h | refine> [variants] [rounding] [inexact] [kbits]"""
import sys, os, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples, meshgen
from oracle import oracle as orc


def pcg(A, b, Bop, rtol=1e-10, maxit=500):
    x = np.zeros_like(b); r = b.copy(); z = Bop(r); p = z.copy(); rz = r @ z; n0 = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q
        if np.linalg.norm(r) <= rtol * n0: return it
        z = Bop(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit


def main(case, arg, what):
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
    cfix = np.zeros(3 * nc, bool)
    for c in corners:
        for k in range(3):
            if fixed[3 * c + k]: cfix[3 * f2c[c] + k] = True
    P = P[:, np.where(~cfix)[0]].tocsr()
    A1 = (P.T @ K @ P).tocsc(); M = spla.factorized(A1)
    print(f"{case} {arg}: {len(el)} tet10, n={n}, corners {nc}, mids {len(mids)}", flush=True)

    def lam_max(Sop):
        return spla.eigsh(spla.LinearOperator((n, n), matvec=lambda v: Sop(K @ v)), k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0]

    def weight(lm, ratio=16.0, safety=1.02):
        return 2.0 / (safety * lm * (1.0 + 1.0 / ratio))

    def mult(Sop, w, Mc=M, Kq=K):
        def B(r):
            z0 = w * Sop(r); r1 = r - Kq @ z0; z = z0 + P @ Mc(P.T @ r1)
            return z + w * Sop(r - Kq @ z)
        return B

    def dofs(nodes):
        return (3 * np.asarray(nodes)[:, None] + np.arange(3)[None, :]).ravel()

    def schwarz(patches, rounding=None, scale=None):
        R, C, V = [], [], []
        for idx in patches:
            Ki = np.linalg.inv(K[idx][:, idx].toarray())
            if rounding == "fp32": Ki = Ki.astype(np.float32).astype(np.float64)
            elif rounding == "fp16": s_ = np.abs(Ki).max(); Ki = (Ki / s_).astype(np.float16).astype(np.float64) * s_
            elif rounding == "bf16":
                a = Ki.astype(np.float32).view(np.uint32); Ki = ((a + 0x8000) & 0xFFFF0000).astype(np.uint32).view(np.float32).astype(np.float64)
            elif rounding == "i16": s_ = np.abs(Ki).max(); Ki = np.round(Ki / s_ * 32767) / 32767 * s_
            if scale is not None: d = scale[idx]; Ki = d[:, None] * Ki * d[None, :]
            rr, cc = np.meshgrid(idx, idx, indexing="ij")
            R.append(rr.ravel()); C.append(cc.ravel()); V.append(Ki.ravel())
        return sp.coo_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(C))), shape=(n, n)).tocsr()

    Dinv = sp.block_diag([np.linalg.inv(K[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray()) for i in range(nn)]).tocsr()
    SJ = lambda r: Dinv @ r
    lmJ = lam_max(SJ)
    print(f"  nodal block-Jacobi: lmax {lmJ:.3f}: {pcg(K, f, mult(SJ, weight(lmJ)))} iterations", flush=True)
    star = [[] for _ in range(nn)]
    for m in mids:
        star[pa[m]].append(m); star[pb[m]].append(m)
    vp = [dofs([c] + star[c]) for c in corners]
    S = schwarz(vp)
    Sv = lambda r: S @ r
    lmV = lam_max(Sv)
    print(f"  vertex patches: {len(vp)}, {np.mean([len(p) for p in vp]):.1f} unknowns each, nnz(S) / nnz(K) = {S.nnz / K.nnz:.2f}, lmax {lmV:.3f}: "
          f"{pcg(K, f, mult(Sv, weight(lmV)))} iterations", flush=True)
    if "variants" in what:
        S2 = schwarz([dofs([m, pa[m], pb[m]]) for m in mids]); l2 = lam_max(lambda r: S2 @ r)
        print(f"  edge patches (mid-side node + its corners): nnz(S) / nnz(K) = {S2.nnz / K.nnz:.2f}, lmax {l2:.3f}: {pcg(K, f, mult(lambda r: S2 @ r, weight(l2)))} iterations", flush=True)
        cnt = np.zeros(n)
        for idx in vp: cnt[idx] += 1
        Sw = schwarz(vp, scale=1.0 / np.sqrt(cnt)); lw = lam_max(lambda r: Sw @ r)
        print(f"  vertex patches, partition-of-unity scaled: lmax {lw:.3f}: {pcg(K, f, mult(lambda r: Sw @ r, weight(lw)))} iterations", flush=True)
        S3 = schwarz([dofs(star[c]) for c in corners if len(star[c])])
        cd = dofs(corners); Dc = sp.coo_matrix(sp.block_diag([np.linalg.inv(K[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray()) for i in corners]))
        S3 = (S3 + sp.coo_matrix((Dc.data, (cd[Dc.row], cd[Dc.col])), shape=(n, n))).tocsr(); l3 = lam_max(lambda r: S3 @ r)
        print(f"  edge stars + corner Jacobi: lmax {l3:.3f}: {pcg(K, f, mult(lambda r: S3 @ r, weight(l3)))} iterations", flush=True)
    if "rounding" in what:
        for rd in ("fp32", "fp16", "bf16", "i16"):
            Sr = schwarz(vp, rounding=rd)
            print(f"  vertex patches, inverses rounded to {rd}: {pcg(K, f, mult(lambda r: Sr @ r, weight(lmV)))} iterations", flush=True)
    if "inexact" in what:
        n1 = A1.shape[0]; A1r = A1.tocsr(); d1 = 1.0 / A1r.diagonal()
        l1 = 1.02 * spla.eigsh(spla.LinearOperator((n1, n1), matvec=lambda v: d1 * (A1r @ v)), k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0]
        def cheb_solver(k, alpha):
            lmin = l1 / alpha; theta = 0.5 * (l1 + lmin); delta = 0.5 * (l1 - lmin); sigma = theta / delta
            def solve(b):
                x = np.zeros_like(b); r = b.copy(); d = (d1 * r) / theta; rho = 1.0 / sigma
                for _ in range(k):
                    x += d; r -= A1r @ d
                    rn = 1.0 / (2.0 * sigma - rho)
                    d = rn * rho * d + 2.0 * rn / delta * (d1 * r); rho = rn
                return x
            return solve
        for k, alpha in ((60, 1e3), (200, 1e4)):
            Mc = cheb_solver(k, alpha)
            print(f"  coarse solve = Chebyshev({k}) of point Jacobi on [lmax / {alpha:g}, lmax]: block-Jacobi {pcg(K, f, mult(SJ, weight(lmJ), Mc))}, "
                  f"vertex patches {pcg(K, f, mult(Sv, weight(lmV), Mc))} iterations", flush=True)
    if "kbits" in what:
        Bk = K.tobsr(blocksize=(3, 3))
        for bits in (16, 12, 10, 8):
            Bq = Bk.copy(); s_ = np.abs(Bq.data).max(axis=(1, 2), keepdims=True); s_[s_ == 0] = 1; q = 2 ** (bits - 1) - 1
            Bq.data = np.round(Bq.data / s_ * q) / q * s_
            print(f"  block-Jacobi, smoother matrix rounded to {bits} bits per 3 x 3 block: {pcg(K, f, mult(SJ, weight(lmJ), M, Bq.tocsr()))} iterations", flush=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
