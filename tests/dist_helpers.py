"""Oracle-backed LocalOps for the CPU (gloo) tests of fcvm_hip.dist -- test infrastructure."""
import numpy as np
import scipy.sparse as sp

from oracle import oracle as orc


class OracleLocalOps:
    def __init__(self, part, nocoord_local, material, fix_local, element_offset=0):
        self.el = part.elNodes
        self.xyz = np.ascontiguousarray(nocoord_local)
        self.mat = np.asarray(material, dtype=np.float64)
        self.fix = dict(fix_local)
        self.element_offset = element_offset
        self.K = None
        self.dinv = None

    def assemble(self, grav, disp, sig, pgp, Et_E, glv, modf, xgp):
        d = None if disp is None else disp.get()
        s = None if sig is None else sig.get()
        pg = None if pgp is None else pgp.get()
        stm, row, col, g, m, V, *_r, x = orc.calcGSM(self.el, self.xyz, self.mat, self.fix, grav, orc.Loads(), disp=d,
                                                     sig=s, pgp=pg, Et_E=Et_E)
        n = 3 * len(self.xyz)
        low = sp.coo_matrix((stm, (row, col)), shape=(n, n)).tocsr()
        self.K = (low + sp.tril(low, -1).T).tocsr()
        glv.set(g)
        modf.set(m)
        if xgp is not None:
            xgp.set(x.reshape(-1))
        return V

    def get_diag_blocks(self, dblk):
        nn = len(self.xyz)
        out = np.zeros((nn, 3, 3))
        for i in range(nn):
            out[i] = self.K[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray()
        dblk.set(out.reshape(-1))

    def set_diag_blocks(self, dblk):
        self.dinv = np.linalg.inv(dblk.get().reshape(-1, 3, 3))

    def update_stress_load(self, sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, Et_E, LD, pgp):
        ne, nn = len(self.el), len(self.xyz)
        sn, st, q = np.zeros(24 * ne), np.zeros(24 * ne), np.zeros(3 * nn)
        pg = np.zeros(4 * ne, np.uint8)
        dn = np.zeros(3 * nn) if disp_new is None else disp_new.get()
        orc.update_stress_load(self.el, self.xyz, self.mat, sig_yield.get(), dn, du.get(), sig_old.get(), sn, st, q, Et_E,
                               LD, pg)
        sig_new.set(sn); sig_test.set(st); qin.set(q); pgp.set(pg)

    def update_peeq_csr(self, sig_test, sig_new, sig_yield, ultimate_strain, peeq, csr, triax, pressure, sigmises, ecr, Et_E):
        a = {k: v.get() for k, v in dict(sy=sig_yield, peeq=peeq, csr=csr, triax=triax, pressure=pressure,
                                         sigmises=sigmises, ecr=ecr).items()}
        orc.update_PEEQ_CSR(len(self.el), self.mat, sig_test.get(), sig_new.get(), a["sy"], ultimate_strain, a["peeq"],
                            a["csr"], a["triax"], a["pressure"], a["sigmises"], a["ecr"], Et_E)
        for k, v in dict(sy=sig_yield, peeq=peeq, csr=csr, triax=triax, pressure=pressure, sigmises=sigmises, ecr=ecr).items():
            v.set(a[k])
        ml = int(np.argmax(a["csr"]))
        return dict(maxloc=ml, csr_max=a["csr"][ml], peeq_max=a["peeq"].max(), pressure=a["pressure"][ml],
                    sigmises=a["sigmises"][ml], triax=a["triax"][ml], ecr=a["ecr"][ml], peeq=a["peeq"][ml])

    def spmv(self, x, y):
        y.set(self.K @ x.get())

    # -- buckling pre-analysis: K_b + sigma K_g and K_g of the rank's elements (oracle element matrices) --------------
    def assemble_buckling(self, sig, sigma):
        ne, n = len(self.el), 3 * len(self.xyz)
        isfix, _ = orc.fix_arrays(self.fix, n)
        r, c = np.zeros(900 * ne, np.int64), np.zeros(900 * ne, np.int64)
        ks, kg = np.zeros(900 * ne), np.zeros(900 * ne)
        orc.lib().orc_buckling_matrices(ne, self.el, self.xyz, self.mat[0], self.mat[1], isfix, sig.get(), r, c, ks, kg)
        self.Kg = sp.csr_matrix((kg, (r, c)), shape=(n, n))
        self.K = (sp.csr_matrix((ks, (r, c)), shape=(n, n)) + sigma * self.Kg).tocsr()

    def spmv_geometric(self, x, y):
        y.set(self.Kg @ x.get())

    def set_coords(self, nocoord_local):
        self.xyz = np.ascontiguousarray(nocoord_local, dtype=np.float64)

    # -- multigrid building blocks, numpy / scipy restatement for the CPU tests -----------------
    def mg_enable(self):
        self.mg = True

    def _hier(self):
        if getattr(self, "_hier_K", None) is self.K:
            return
        el0 = self.el - 1
        nn = len(self.xyz)
        corners = np.unique(el0[:, :4])
        f2c = np.full(nn, -1)
        f2c[corners] = np.arange(len(corners))
        pa, pb = f2c.copy(), f2c.copy()
        pairs = ((0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3))
        for j, (a, b) in enumerate(pairs):
            pa[el0[:, 4 + j]] = f2c[el0[:, a]]
            pb[el0[:, 4 + j]] = f2c[el0[:, b]]
        nc = len(corners)
        rows, cols, vals = [], [], []
        for i in range(nn):
            if pa[i] < 0:
                continue
            for k in range(3):
                rows += [3 * i + k, 3 * i + k]
                cols += [3 * pa[i] + k, 3 * pb[i] + k]
                vals += [0.5, 0.5]
        P = sp.csr_matrix((vals, (rows, cols)), shape=(3 * nn, 3 * nc))
        free0 = np.ones(3 * nn)
        free0[list(self.fix.keys())] = 0.0
        free1 = free0.reshape(-1, 3)[corners].reshape(-1)
        self.Pm = sp.diags(free0) @ P @ sp.diags(free1)
        self.A1 = (self.Pm.T @ self.K @ self.Pm + sp.diags(1.0 - free1)).tocsr()
        self.c2f, self.free1 = corners, free1
        self.dinv1 = None
        self._hier_K = self.K

    def mg_coarse_map(self):
        self._hier()
        return self.c2f.astype(np.int32)

    def mg_coarse_nodes(self):
        self._hier()
        f = (1.0 - self.free1).reshape(-1, 3).astype(np.uint8)
        return self.xyz[self.c2f], (f[:, 0] | (f[:, 1] << 1) | (f[:, 2] << 2)).astype(np.uint8)

    def _A(self, level):
        self._hier()
        return self.K if level == 0 else self.A1

    def mg_spmv(self, level, x, y):
        y.set(self._A(level) @ x.get())

    def spmv_lo(self, x, y):
        """the smoother's product of the multiplicative cycle (the HIP path streams a 16-bit copy; the exact matrix here)"""
        y.set(self.K @ x.get())

    def mg_get_diag(self, level, dblk):
        A = self._A(level)
        n = A.shape[0] // 3
        out = np.zeros((n, 3, 3))
        for i in range(n):
            out[i] = A[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray()
        dblk.set(out.reshape(-1))

    def mg_set_diag(self, level, dblk):
        inv = np.linalg.inv(dblk.get().reshape(-1, 3, 3))
        if level == 0:
            self.dinv = inv
        else:
            self.dinv1 = inv

    def _dinv(self, level, v):
        d = self.dinv if level == 0 else self.dinv1
        return np.einsum("nij,nj->ni", d, v.reshape(-1, 3)).reshape(-1)

    def mg_cheb_init(self, level, b, x, r, d, c0):
        bb = b.get()
        x.set(np.zeros_like(bb)); r.set(bb); d.set(c0 * self._dinv(level, bb))

    def mg_cheb_step(self, level, t, x, r, d, cdd, cdr, update_d):
        dd = d.get()
        x.set(x.get() + dd)
        if t is not None:
            rr = r.get() - t.get()
            r.set(rr)
            if update_d:
                d.set(cdd * dd + cdr * self._dinv(level, rr))

    def mg_restrict(self, r0, r1):
        self._hier()
        r1.set(self.Pm.T @ r0.get())

    def mg_restrict_weighted(self, r0, w, r1):
        self._hier()
        r1.set(self.Pm.T @ (w.get() * r0.get()))

    def mg_prolong_add(self, e1, z0, z):
        self._hier()
        z.set(z0.get() + self.Pm @ e1.get())

    def pcg_update_xr_z0(self, alpha, p, q, x, r, z0, weight):
        a = float(alpha.item())
        x.set(x.get() + a * p.get())
        rn = r.get() - a * q.get()
        r.set(rn)
        z0.set(weight * self._dinv(0, rn))

    def mg_set_level2(self, nagg, agg, Z):
        nc = len(agg)
        ii = np.repeat(np.arange(nc), 18)
        cc = np.tile(np.repeat(np.arange(3), 6), nc)
        kk = np.tile(np.arange(6), 3 * nc)
        self.Z2 = sp.csr_matrix((Z.reshape(-1), (3 * ii + cc, 6 * agg[ii].astype(np.int64) + kk)), shape=(3 * nc, 6 * nagg))
        self.nagg = nagg

    def mg_galerkin_blocks(self):
        """this rank's share of Z^T A1 Z as 6 x 6 blocks over the global aggregates (HIP path: k_galerkin_blocks)"""
        self._hier()
        nagg = self.nagg
        A2 = (self.Z2.T @ (self.A1 @ self.Z2)).tocoo()
        # the block pattern follows the aggregate graph of A1 (not the numerical non-zeros, as the device kernel's lists do)
        G = sp.csr_matrix((np.ones(self.Z2.nnz), (self.Z2.tocoo().row // 3, self.Z2.tocoo().col // 6)), shape=(self.A1.shape[0] // 3, nagg))
        A1g = sp.csr_matrix((np.ones(self.A1.nnz), (self.A1.tocoo().row // 3, self.A1.tocoo().col // 3)), shape=(G.shape[0], G.shape[0]))
        pat = (G.T @ A1g @ G).tocoo()
        keys = np.unique(pat.row.astype(np.int64) * nagg + pat.col)
        B = np.zeros((len(keys), 36))
        pos = np.searchsorted(keys, (A2.row // 6).astype(np.int64) * nagg + A2.col // 6)
        np.add.at(B, (pos, (A2.row % 6) * 6 + A2.col % 6), A2.data)
        sptr = np.zeros(nagg + 1, np.int64)
        np.add.at(sptr, keys // nagg + 1, 1)
        return np.cumsum(sptr), (keys % nagg).astype(np.int32), B.reshape(-1)

    def mg_l2_finish(self, nagg, arank, sptr_g, slot_agg, B, nsuper, sagg, soff, Cm):
        """numpy restatement of fcvm_mg_l2_finish: the replicated level from the summed blocks -- dense inverse, or the sparse cycle of
        csrc/fcvm_l2s.inc (block-Jacobi Chebyshev sweeps around dense super-aggregate corrections)"""
        from fcvm_hip.dist import _Cheb
        rows = np.repeat(np.arange(nagg), np.diff(sptr_g))
        Bb = B.reshape(-1, 6, 6)
        r = (6 * rows[:, None, None] + np.arange(6)[None, :, None]) + np.zeros((1, 1, 6), np.int64)
        c = (6 * slot_agg.astype(np.int64)[:, None, None] + np.arange(6)[None, None, :]) + np.zeros((1, 6, 1), np.int64)
        A2 = sp.csr_matrix((Bb.reshape(-1), (r.reshape(-1), c.reshape(-1))), shape=(6 * nagg, 6 * nagg))
        A2 = (0.5 * (A2 + A2.T)).tolil()
        for g in range(nagg):
            for k in range(int(arank[g]), 6):
                A2[6 * g + k, 6 * g + k] = 1.0
        A2 = A2.tocsr()
        self.l2s = None
        if not nsuper:
            Ad = A2.toarray()
            self.A2inv = np.linalg.inv(Ad + 1e-13 * np.trace(Ad) / len(Ad) * np.eye(len(Ad)))
            return
        dinv = np.stack([np.linalg.inv(A2[6 * g:6 * g + 6, 6 * g:6 * g + 6].toarray()) for g in range(nagg)])
        n3 = int(soff[-1])
        rr, cc, vv = [], [], []
        for g in range(nagg):
            S = int(sagg[g])
            for m in range(int(soff[S + 1] - soff[S])):
                for k in range(6):
                    rr.append(6 * g + k); cc.append(int(soff[S]) + m); vv.append(Cm[g, k, m])
        C = sp.csr_matrix((vv, (rr, cc)), shape=(6 * nagg, n3))
        A3 = (C.T @ A2 @ C).toarray()
        A3inv = np.linalg.inv(A3 + 1e-13 * np.trace(A3) / n3 * np.eye(n3))
        apply_dinv = lambda v: np.einsum("gkl,gl->gk", dinv, v.reshape(-1, 6)).reshape(-1)  # noqa: E731
        x = np.random.default_rng(5).standard_normal(6 * nagg)
        lam = 0.0
        for _ in range(100):
            y = apply_dinv(A2 @ x)
            lam = np.linalg.norm(y) / np.linalg.norm(x)
            x = y / np.linalg.norm(y)
        self.l2s = dict(A2=A2, dinv=apply_dinv, C=C, A3inv=A3inv, ch=_Cheb(1.15 * lam, 100.0, 8), k=8, ncoarse=2)

    def _l2s_cycle(self, b):
        L = self.l2s
        A2, dinv, ch, k = L["A2"], L["dinv"], L["ch"], L["k"]
        st = dict(x=np.zeros_like(b), r=b.copy(), d=ch.c0 * dinv(b))

        def step(cdd, cdr, update, add):
            t = A2 @ st["d"]
            st["x"] += st["d"]
            st["r"] -= t
            if update:
                st["d"] = cdd * st["d"] + cdr * dinv(st["r"])
                if add:
                    st["x"] += st["d"]

        def coarse():
            st["d"] = L["C"] @ (L["A3inv"] @ (L["C"].T @ st["r"]))
        for j in range(1, k):
            step(ch.cdd[j], ch.cdr[j], True, False)
        step(0.0, 0.0, False, False)
        coarse()
        for _ in range(1, L["ncoarse"]):
            step(0.0, ch.c0, True, False)
            for j in range(1, k):
                step(ch.cdd[j], ch.cdr[j], True, False)
            step(0.0, 0.0, False, False)
            coarse()
        step(0.0, ch.c0, True, k == 1)
        for j in range(1, k):
            step(ch.cdd[j], ch.cdr[j], True, j == k - 1)
        return st["x"]

    def mg_l2_restrict(self, r1, r2):
        r2.set(self.Z2.T @ r1.get())

    def mg_l2_solve(self, r2, e2):
        e2.set(self._l2s_cycle(r2.get()) if self.l2s is not None else self.A2inv @ r2.get())

    def mg_l2_prolong(self, e2, d1):
        d1.set(self.Z2 @ e2.get())

    def mg_prolong(self, e1, d0):
        self._hier()
        d0.set(self.Pm @ e1.get())

    def precond(self, r, z):
        z.set(np.einsum("nij,nj->ni", self.dinv, r.get().reshape(-1, 3)).reshape(-1))


def split_case(case, rank, world):
    """Range partition of the elements of a meshgen.cube_case; returns the rank's local pieces."""
    from fcvm_hip import meshgen
    el, xyz = case["elNodes"], case["nocoord"]
    ne = len(el)
    lo, hi = rank * ne // world, (rank + 1) * ne // world
    return el[lo:hi], lo, meshgen
