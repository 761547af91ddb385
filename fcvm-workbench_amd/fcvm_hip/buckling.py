"""Elastic (linear) buckling pre-analysis of the reference's "GNLY" mode on the GPU.

Reference: fcVM.py:1199-1214 -- K, G of calcTSM(nstep = 1) and
``eigsh(K, k=2, M=G, sigma=0.1, which='LM', mode='buckling')``, i.e. the two buckling factors lambda
of K x = lambda G x closest to sigma = 0.1 -- and fcVM.py:1224-1240, the imperfection drawn from the
two buckling shapes.

ARPACK's shift-invert Lanczos is replaced by block subspace iteration with Rayleigh-Ritz on the same
shifted operator: theta = 1/(lambda - sigma) are the dominant eigenvalues of (K - sigma G)^-1 G; the
inverse is applied with the multigrid-PCG solver of libfcvm_hip on A = K_b + sigma K_g (the matrices
are assembled by fcvm_assemble_buckling; G = -K_g).  All vectors stay on the device; the host only
sees the m x m Ritz problem.
"""
import numpy as np
import scipy.linalg as sla


def buckling_factors(model, grav, k=2, sigma=0.1, block=6, tol=1e-10, maxit=80, rtol_solve=1e-11, log=None):
    """Returns (eigenvalues [k], eigenvectors host array [ndof, k], ue device vector of the elastic solution).
    Selection as ARPACK's buckling mode makes it (eigsh(K, M=G, sigma, which='LM', mode='buckling'), fcVM.py:1208):
    largest |nu|, nu = lambda / (lambda - sigma) = 1 + sigma * theta with theta = 1 / (lambda - sigma) the Ritz
    values of (K - sigma G)^-1 G -- i.e. the smallest factors ABOVE sigma first (nu > 1); negative factors
    (load reversal, nu < 1) only when there are not enough positive ones.  The block grows until it holds k
    positive Ritz values (tension-dominated models have negative factors closer to zero than the wanted ones).
    Leaves the context with the buckling matrices assembled: call model.calc_gsm() again before the next
    ordinary solve."""
    c = model.ctx
    say = log if log is not None else (lambda *a: None)
    ndof, ne = model.ndof, model.ne
    glv, modf, _, _ = model.calc_gsm(grav, want_x=False)
    f = c.mul(model.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    ue, _ = c.solve(f, rtol=rtol_solve)
    # elastic stresses of ue at the full load (fcVM.py:1195-1197)
    sig, sig_t = c.empty(24 * ne), c.empty(24 * ne)
    c.update_stress_load(c.array(np.full(4 * ne, 1.0e30)), None, ue, c.zeros(24 * ne), sig, sig_t, c.empty(ndof), 0.0, False,
                         c.empty(4 * ne, np.uint8))
    c.assemble_buckling(sig, sigma)
    m = max(block, k + 2)
    nstart = [0]

    def start_vector():   # the same on one GPU and on any partition: a fixed function of the global dof number
        nstart[0] += 1
        return c.start_vector(nstart[0])

    X = [start_vector() for _ in range(m)]
    Y = [c.empty(ndof) for _ in range(m)]
    tmp = c.empty(ndof)

    def G(x, y):  # y = G x = -K_g x
        c.spmv_geometric(x, tmp)
        c.axpby(-1.0, tmp, 0.0, None, out=y)

    theta_old = None
    lam = None
    for it in range(maxit):
        for j in range(m):
            G(X[j], Y[j])
        # Z = A^-1 G X,  so A Z = Y: the block's systems two at a time, every matrix pass shared (Context.solve2)
        Z = []
        for j in range(0, m, 2):
            if j + 1 < m and hasattr(c, "solve2"):
                Z += list(c.solve2(Y[j], Y[j + 1], rtol=(rtol_solve, rtol_solve))[:2])
            else:
                Z += [c.solve(Y[jj], rtol=rtol_solve)[0] for jj in range(j, min(j + 2, m))]
        GZ = []
        for j in range(m):
            g = c.empty(ndof)
            G(Z[j], g)
            GZ.append(g)
        Ga = np.array([[c.dot(Z[i], GZ[j]) for j in range(m)] for i in range(m)])
        Aa = np.array([[c.dot(Z[i], Y[j]) for j in range(m)] for i in range(m)])
        Ga, Aa = 0.5 * (Ga + Ga.T), 0.5 * (Aa + Aa.T)
        th, C = sla.eigh(Ga, Aa)  # Ga c = theta Aa c, Aa SPD (A is SPD below the first buckling load)
        order = np.argsort(-np.abs(1.0 + sigma * th), kind="stable")
        th, C = th[order], C[:, order]
        Xn = []
        for j in range(m):  # X_j = sum_i C[i,j] Z_i, scaled to unit max-norm-ish via Aa-normalisation of eigh
            x = c.axpby(C[0, j], Z[0], 0.0, None)
            for i in range(1, m):
                c.axpby(1.0, x, C[i, j], Z[i], out=x)
            Xn.append(x)
        X = Xn
        lam = sigma + 1.0 / th[:k]
        say("buckling subspace iteration %d: factors %s" % (it, lam))
        if theta_old is not None and len(theta_old) == k and np.all(np.abs(th[:k] - theta_old) <= tol * np.abs(th[:k])):
            if np.all(th[:k] > 0.0) or m >= 4 * block:
                break
            # converged onto a block without k positive factors: widen it and go on
            extra = [start_vector() for _ in range(block)]
            X, Y, m = X + extra, Y + [c.empty(ndof) for _ in range(block)], m + block
            say("buckling subspace: block widened to %d (negative factors nearer to the shift than the wanted ones)" % m)
            theta_old = None
            continue
        theta_old = th[:k].copy()
    # ARPACK's buckling mode returns K-normalised vectors (x^T K x = 1; checked against scipy); the Ritz vectors
    # are A-normalised (A = K - sigma G) with x^T G x = theta, so x^T K x = 1 + sigma * theta.  The relative scale of
    # the two shapes matters for the imperfection mix of fcVM.py:1226-1227.
    vec = np.stack([X[j].get() / np.sqrt(1.0 + sigma * th[j]) for j in range(k)], axis=1)
    return lam, vec, ue


def imperfection(eigenvec, ev1, ev2, maxImp, global_extreme=None):
    """fcVM.py:1226-1236: imperfection shape from the two buckling modes, scaled to maxImp.
    global_extreme(v) -> (max |v|, sign of that entry) over all ranks of a partitioned model (None: this process holds
    the whole vector)."""
    def extreme(v):
        if global_extreme is not None:
            return global_extreme(v)
        i = np.argmax(np.abs(v))
        return float(np.abs(v[i])), float(np.sign(v[i]))
    ua = ev1 / (ev1 + ev2) * eigenvec[:, 0] + ev2 / (ev1 + ev2) * eigenvec[:, 1]
    ub = ev1 / (ev1 + ev2) * eigenvec[:, 0] - ev2 / (ev1 + ev2) * eigenvec[:, 1]
    (ma, sa), (mb, sb) = extreme(ua), extreme(ub)
    if ma > mb:
        return maxImp / ma * sa * ua
    return maxImp / mb * sb * ub
