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

    def precond(self, r, z):
        z.set(np.einsum("nij,nj->ni", self.dinv, r.get().reshape(-1, 3)).reshape(-1))


def split_case(case, rank, world):
    """Range partition of the elements of a meshgen.cube_case; returns the rank's local pieces."""
    from fcvm_hip import meshgen
    el, xyz = case["elNodes"], case["nocoord"]
    ne = len(el)
    lo, hi = rank * ne // world, (rank + 1) * ne // world
    return el[lo:hi], lo, meshgen
