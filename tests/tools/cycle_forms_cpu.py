#!/usr/bin/env python3
"""CPU prototype (scipy, oracle-assembled K): PCG iteration counts of two-level cycle FORMS on the tet10 -> tet4 hierarchy with an
exact coarse solve, to see what a form is worth before it is written in HIP.  (Test infrastructure: uses oracle/.)
  B_mult = S + S - S K S + W M W^T,  W = (I - S K) P     the multiplicative V(1,1) cycle of the library (two K passes)
  B_add  = S + P M P^T                                    additive
  B_sp   = S + W M W^T                                    additive with the smoothed transfer W (only products with G = K P)
  B_sp2  = S + W M2 W^T, M2 = (W^T K W)^-1                smoothed transfer with its own Galerkin operator
and, inside the multiplicative form, other fine-level smoothers (Chebyshev(2) block-Jacobi, block Gauss-Seidel forward / backward).
Round-3 result (plate, 3840 / 16830 tet10, exact coarse solve): mult 44 / 38, add 99 / 84, sp 85 / 72, sp2 93 / 78 iterations;
Chebyshev(2) 29, Gauss-Seidel 31 at 3840 tet10 -- with the library's approximate level-1 solve the second smoothing step brings
7 % (gpurun_out/r03ay), not 34 %: see DESIGN.md "Linear solver".
"""
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

def main(h):
    pc = examples.plate_with_hole(h)
    el, xyz = np.asarray(pc["elNodes"]), np.asarray(pc["nocoord"])
    nn = len(xyz); n = 3 * nn
    stm, row, col, glv, modf, *_ = orc.calcGSM(el, xyz, pc["material"], pc["fix"], (0.0, 0.0, 0.0), orc.Loads(ufaces=pc["ufaces"], uloads=pc["uloads"]))
    low = sp.coo_matrix((stm, (row, col)), shape=(n, n)).tocsr()
    K = (low + sp.tril(low, -1).T).tocsr()
    fixed = np.zeros(n, bool); fixed[list(pc["fix"].keys())] = True
    f = np.where(fixed, 0.0, glv) + modf
    e0 = el - 1
    corners = np.unique(e0[:, :4]); f2c = -np.ones(nn, np.int64); f2c[corners] = np.arange(len(corners)); nc = len(corners)
    rows, cols, vals = [], [], []
    for c in corners: rows.append(c); cols.append(f2c[c]); vals.append(1.0)
    pa, pb = -np.ones(nn, np.int64), -np.ones(nn, np.int64)
    for k, (a, b) in enumerate(meshgen.MID_PAIRS):
        pa[e0[:, 4 + k]] = f2c[e0[:, a]]; pb[e0[:, 4 + k]] = f2c[e0[:, b]]
    for m in np.where(pa >= 0)[0]:
        rows += [m, m]; cols += [pa[m], pb[m]]; vals += [0.5, 0.5]
    Pn = sp.coo_matrix((vals, (rows, cols)), shape=(nn, nc)).tocsr()
    P = sp.kron(Pn, sp.identity(3)).tocsr()
    # supported dofs: no correction there (rows of P zeroed), coarse dofs of supported corner dofs removed
    free = ~fixed
    P = sp.diags(free.astype(float)) @ P
    cfix = np.zeros(3 * nc, bool)
    for c in corners:
        for k in range(3):
            if fixed[3 * c + k]: cfix[3 * f2c[c] + k] = True
    keep = np.where(~cfix)[0]
    P = P[:, keep].tocsr()
    # block-Jacobi
    D = sp.block_diag([K[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray() for i in range(nn)]).tocsr()
    Dinv = sp.block_diag([np.linalg.inv(K[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray()) for i in range(nn)]).tocsr()
    lmax = spla.eigsh(spla.LinearOperator((n, n), matvec=lambda v: Dinv @ (K @ v)), k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0]
    hi = 1.1 * lmax; lo = hi / 16.0; c0 = 2.0 / (hi + lo)
    S = c0 * Dinv
    A1 = (P.T @ K @ P).tocsc(); M = spla.factorized(A1)
    W = (P - S @ (K @ P)).tocsr()
    A1w = (W.T @ K @ W).tocsc(); M2 = spla.factorized(A1w)
    def B_mult(r):
        z0 = S @ r; r1 = r - K @ z0; e = M(P.T @ r1); z = z0 + P @ e
        return z + S @ (r - K @ z)
    def B_add(r): return S @ r + P @ M(P.T @ r)
    def B_sp(r): return S @ r + W @ M(W.T @ r)
    def B_sp2(r): return S @ r + W @ M2(W.T @ r)
    def B_sp_w(w):
        return lambda r: w * (S @ r) + W @ M(W.T @ r)
    # ---- other fine-level smoothers inside the multiplicative form (pre: forward, post: its adjoint) ----
    def make_mult(pre, post):
        def B(r):
            z0 = pre(r, None); r1 = r - K @ z0; e = M(P.T @ r1); z = z0 + P @ e
            return post(r, z)
        return B
    # Chebyshev(2) Jacobi on [hi/16, hi]
    th, de = 0.5 * (hi + lo), 0.5 * (hi - lo)
    def cheb2(r, z):
        z = np.zeros_like(r) if z is None else z
        res = r - K @ z if z.any() else r.copy()
        sig = th / de; rho = 1.0 / sig
        d = (Dinv @ res) / th; z = z + d
        res = res - K @ d
        rn = 1.0 / (2.0 * sig - rho)
        d = rn * rho * d + (2.0 * rn / de) * (Dinv @ res)
        return z + d
    # exact block Gauss-Seidel over node blocks in Morton order (forward / backward)
    q = ((xyz - xyz.min(0)) / (xyz.max(0) - xyz.min(0)).max() * 1023).astype(np.int64)
    def spread(v):
        out = np.zeros_like(v)
        for b in range(10): out |= ((v >> b) & 1) << (3 * b)
        return out
    order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")
    perm = (3 * order[:, None] + np.arange(3)).ravel()
    Kp = K[perm][:, perm].tocsr()
    def split(chunk_of_node):
        # block lower (earlier chunks), block diagonal part (same chunk: only the 3x3 node blocks are inverted: Jacobi within a chunk)
        cd = np.repeat(chunk_of_node, 3)
        coo = Kp.tocoo()
        lowm = sp.coo_matrix((coo.data * (cd[coo.row] > cd[coo.col]), (coo.row, coo.col)), shape=Kp.shape).tocsr(); lowm.eliminate_zeros()
        return lowm
    Dp = sp.block_diag([Kp[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray() for i in range(nn)]).tocsr()
    def gs_pair(chunk_of_node, damp):
        Lm = split(chunk_of_node)
        Fw = spla.splu((Dp / damp + Lm).tocsc())          # (D/w + L) forward; its transpose backward
        Bw = spla.splu((Dp / damp + Lm.T).tocsc())
        def pre(r, z):   # from zero
            return unperm(Fw.solve(r[perm]))
        def post(r, z):
            res = (r - K @ z)[perm]
            return z + unperm(Bw.solve(res))
        return pre, post
    inv = np.empty_like(perm); inv[perm] = np.arange(n)
    def unperm(v): return v[inv]
    node_rank = np.arange(nn)
    variants = [("jacobi V(1,1) (library)", B_mult), ("chebyshev(2) jacobi", make_mult(cheb2, cheb2))]
    pre, post = gs_pair(node_rank, 1.0)
    variants.append(("block Gauss-Seidel over nodes (Morton order)", make_mult(pre, post)))
    # round 4: "hybrid" block Gauss-Seidel -- Gauss-Seidel over the nodes INSIDE a group of consecutive Morton nodes (what a workgroup of
    # the window format could do in LDS, its rows colour by colour), damped Jacobi BETWEEN the groups (no dependency between
    # workgroups): splitting matrix Q = D / w + (couplings to earlier nodes of the same group); pre forward, post its adjoint
    def hybrid_pair(gsize, damp):
        grp = np.arange(nn) // gsize                 # (nodes are in Morton order after the permutation)
        cd = np.repeat(grp, 3)
        nd3 = np.repeat(np.arange(nn), 3)
        coo = Kp.tocoo()
        m = (cd[coo.row] == cd[coo.col]) & (nd3[coo.row] > nd3[coo.col])
        Lm = sp.coo_matrix((coo.data * m, (coo.row, coo.col)), shape=Kp.shape).tocsr(); Lm.eliminate_zeros()
        Fw = spla.splu((Dp / damp + Lm).tocsc()); Bw = spla.splu((Dp / damp + Lm.T).tocsc())
        def pre(r, z): return unperm(Fw.solve(r[perm]))
        def post(r, z): return z + unperm(Bw.solve((r - K @ z)[perm]))
        return pre, post
    # ... and its GPU-shaped form: the rows of a group dealt to `nclass` classes (the waves of a workgroup, no spatial meaning),
    # Gauss-Seidel from class to class, Jacobi inside a class
    def class_pair(gsize, nclass, damp):
        grp = np.arange(nn) // gsize
        cls = np.random.default_rng(1).integers(0, nclass, nn)
        cd, cl3 = np.repeat(grp, 3), np.repeat(cls, 3)
        coo = Kp.tocoo()
        m = (cd[coo.row] == cd[coo.col]) & (cl3[coo.row] > cl3[coo.col])
        Lm = sp.coo_matrix((coo.data * m, (coo.row, coo.col)), shape=Kp.shape).tocsr(); Lm.eliminate_zeros()
        Fw = spla.splu((Dp / damp + Lm).tocsc()); Bw = spla.splu((Dp / damp + Lm.T).tocsc())
        def pre(r, z): return unperm(Fw.solve(r[perm]))
        def post(r, z): return z + unperm(Bw.solve((r - K @ z)[perm]))
        return pre, post
    if os.environ.get("CYCLE_FORMS_COLOURS"):
        # round 5: Gauss-Seidel BETWEEN colours of whole groups (group = 512 consecutive Morton nodes = a workgroup of the window format; colour = group
        # number mod ncol), damped Jacobi inside a colour: on the GPU ncol launches of the 16-bit pass over the colour's group list per smoothing step
        # (the same matrix bytes as one pass), no dependency inside a launch
        for ncol in (2, 4):
            for damp in (c0 * lmax / lmax, 0.65, 0.8, 1.0):
                grp_colour = (np.arange(nn) // 512) % ncol
                pre, post = gs_pair(grp_colour, damp)
                variants.append(("group-colour GS, %d colours of 512-node groups, damping %.2f" % (ncol, damp), make_mult(pre, post)))
    if os.environ.get("CYCLE_FORMS_HYBRID"):
        for nclass in (2, 4, 8):
            for damp in (0.8, 0.7):
                pre, post = class_pair(512, nclass, damp)
                variants.append(("class GS, groups of 512, %d classes, damping %.2f" % (nclass, damp), make_mult(pre, post)))
        for gsize in (512,):
            for damp in (1.0, 0.8, 0.65):
                pre, post = hybrid_pair(gsize, damp)
                variants.append(("hybrid GS, groups of %d nodes, damping %.2f" % (gsize, damp), make_mult(pre, post)))
    for name, B in variants:
        t0 = time.time(); it = pcg(K, f, B); print("  mult, smoother %-40s %4d iterations  (%.1f s)" % (name, it, time.time() - t0), flush=True)
    if os.environ.get("CYCLE_FORMS_DAMPING"):
        # node-type-dependent damping of the block-Jacobi step (corner / mid-side weights) and a scaled coarse correction: all flat
        # or worse than (1, 1) and 1.0 (round 3: 44 -> 46-51 iterations at 3840 tet10)
        iscorner = np.zeros(nn, bool); iscorner[corners] = True
        def run(wc, wm, alpha=1.0):
            w = np.sqrt(np.repeat(np.where(iscorner, wc, wm), 3))
            S0 = sp.diags(w) @ Dinv @ sp.diags(w)
            lm = spla.eigsh(spla.LinearOperator((n, n), matvec=lambda v: S0 @ (K @ v)), k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0]
            h2 = 1.1 * lm; Sw = (2.0 / (h2 + h2 / 16.0)) * S0
            def B(r):
                z0 = Sw @ r; e = alpha * M(P.T @ (r - K @ z0)); z = z0 + P @ e
                return z + Sw @ (r - K @ z)
            return pcg(K, f, B)
        for wc, wm in ((1, 1), (1, 0.7), (1, 1.5), (1, 2.0), (0.7, 1), (1.5, 1), (2, 1)):
            print("  jacobi weights corner %.2f mid-side %.2f: %d iterations" % (wc, wm, run(wc, wm)), flush=True)
        for al in (0.9, 1.1, 1.25):
            print("  coarse correction scaled by %.2f: %d iterations" % (al, run(1, 1, al)), flush=True)
    print("plate h=%g: %d tet10, %d dof, %d coarse dof; nnz blocks K %d, G = K P %d (%.2f x), W %d" %
          (h, len(el), n, len(keep), K.nnz // 9, (K @ P).nnz // 9, (K @ P).nnz / K.nnz, W.nnz // 9))
    for name, B in (("mult", B_mult), ("add", B_add), ("sp", B_sp), ("sp2", B_sp2), ("sp w=2", B_sp_w(2.0)), ("sp w=0.5", B_sp_w(0.5))):
        t0 = time.time(); it = pcg(K, f, B); print("  %-8s %4d iterations  (%.1f s)" % (name, it, time.time() - t0), flush=True)

if __name__ == "__main__":
    for h in [float(v) for v in (sys.argv[1:] or ["10", "6"])]:
        main(h)
