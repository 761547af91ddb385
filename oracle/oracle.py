"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's solver path ("source code/fcVM.py" of
HarryvL/fcVM-workbench): ctypes wrappers over oracle/fcvm_oracle.c for the per-element
routines plus a Python restatement of the Newton-Raphson / Riks driver ``calcDisp``.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Third-party pieces of the reference that are absent from /root/reference and from this image:
  * scikit-sparse ``sksparse.cholmod.cholesky`` (fcVM.py:66-67, 1121, 1130): sparse Cholesky
    factor / solve of the lower-triangular CSC stiffness.  Restated with SciPy's SuperLU
    (``scipy.sparse.linalg.splu``) on the symmetrised matrix -- both are direct solves of the
    same SPD system, equal up to round-off.
  * numba ``@jit`` (fcVM.py:351 ff.): only a compiler for the loops restated in C here.

Parity status: PINNED by tests/test_oracle_golden.py against the reference's own result files:
  * tests/golden/embankment.npz -- GNLN: 30-step gravity-driven elastoplastic collapse analysis, every
    row of the .out table and the f64 nodal fields of the .vtk (7e-15 / 3e-14 relative);
  * tests/golden/barbuckling.npz -- GNLY: the two elastic buckling factors (K / geometric stiffness of
    calcTSM(nstep = 1) + ARPACK buckling mode, fcVM.py:1199-1214) and the geometrically non-linear
    elastic-plastic post-buckling run (calcTSM tangent re-assembly, stress convection, Riks control,
    face-force loads), .out table and f64 nodal fields (2e-9 relative);
  * tests/golden/lug.npz -- Lug_Pressure_Example: pressure load integrals with the element normals (fcVM.py:649-672),
    incremental displacement output and the continuation pass, all 61 rows of the .out and the .vtk fields;
  * tests/golden/u100.npz -- buckling_U100_3D: buckling factors / shapes, imperfection, the three passes of the .out
    (22 rows), the final nodal fields of the .vtk through calcSum over the selection of the shipped .avr;
  * tests/golden/buckling.npz, plate_with_hole.json, vm_uniaxial: see tests/test_oracle_golden.py;
  * principal stresses of four .vtk files.
Not pinned by a reference fixture: edge / vertex load integrals (fcVM.py:673-727), hardening (Et_E > 0) and prescribed
non-zero displacements (no shipped result file with recoverable inputs uses them).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_pd = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_pi = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_pb = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_ps = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libfcvm_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_external_loads.restype = None
        L.orc_external_loads.argtypes = [_pd, C.c_void_p, C.c_int64, _pi, _pd, C.c_int64, _pi, _pd,
                                         C.c_int64, _pi, _pd, C.c_int64, _pi, _pd, _pd]
        L.orc_calc_stiffness.restype = C.c_int64
        L.orc_calc_stiffness.argtypes = [C.c_int64, _pi, _pd, C.c_void_p, C.c_double, C.c_double,
                                         C.c_double, _pb, _pd, C.c_double, C.c_double, C.c_double,
                                         C.c_void_p, C.c_void_p, C.c_double, _pi, _pi, _pd, _pd,
                                         _pd, _pd, C.POINTER(C.c_double)]
        L.orc_update_stress_load.restype = None
        L.orc_update_stress_load.argtypes = [C.c_int64, _pi, _pd, C.c_double, C.c_double, _pd, _pd,
                                             _pd, _pd, _pd, _pd, _pd, C.c_double, C.c_int, _pb]
        L.orc_update_peeq_csr.restype = None
        L.orc_update_peeq_csr.argtypes = [C.c_int64, C.c_double, C.c_double, _pd, _pd, _pd,
                                          C.c_double, _pd, _pd, _pd, _pd, _pd, _pd, C.c_double]
        L.orc_buckling_matrices.restype = None
        L.orc_buckling_matrices.argtypes = [C.c_int64, _pi, _pd, C.c_double, C.c_double, _pb, _pd, _pi, _pi, _pd, _pd]
        L.orc_map_stresses.restype = None
        L.orc_map_stresses.argtypes = [C.c_int, C.c_int64, C.c_int64, _pi, _pd, _pd, _pd, _pd, _ps,
                                       C.c_double, _pd, _pd, _pd, _pd, _pd]
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Loads:
    """Real load entities (the reference's dummy first rows, fcVM.py:260-267, are not stored)."""

    def __init__(self, pfaces=None, pvals=None, vnodes=None, vloads=None, ufaces=None,
                 uloads=None, edges=None, eloads=None):
        z = lambda s, d: np.zeros(s, dtype=d)  # noqa: E731
        self.pfaces = np.ascontiguousarray(pfaces if pfaces is not None else z((0, 6), np.int64), np.int64)
        self.pvals = np.ascontiguousarray(pvals if pvals is not None else z(0, np.float64), np.float64)
        self.vnodes = np.ascontiguousarray(vnodes if vnodes is not None else z(0, np.int64), np.int64)
        self.vloads = np.ascontiguousarray(vloads if vloads is not None else z((0, 3), np.float64), np.float64)
        self.ufaces = np.ascontiguousarray(ufaces if ufaces is not None else z((0, 6), np.int64), np.int64)
        self.uloads = np.ascontiguousarray(uloads if uloads is not None else z((0, 3), np.float64), np.float64)
        self.edges = np.ascontiguousarray(edges if edges is not None else z((0, 3), np.int64), np.int64)
        self.eloads = np.ascontiguousarray(eloads if eloads is not None else z((0, 3), np.float64), np.float64)


def fix_arrays(fix, ndof):
    isfix = np.zeros(ndof, np.uint8)
    fixval = np.zeros(ndof, np.float64)
    for d, v in fix.items():
        isfix[d] = 1
        fixval[d] = v
    return isfix, fixval


def external_loads(nocoord, loads, disp=None):
    glv = np.zeros(nocoord.size, np.float64)
    lib().orc_external_loads(nocoord, _ptr(disp), len(loads.pfaces), loads.pfaces, loads.pvals,
                             len(loads.vnodes), loads.vnodes, loads.vloads, len(loads.ufaces),
                             loads.ufaces, loads.uloads, len(loads.edges), loads.edges, loads.eloads, glv)
    return glv


def calcGSM(elNodes, nocoord, material, fix, grav, loads, disp=None, sig=None, pgp=None, Et_E=0.0):
    """fcVM.py:621-816 calcGSM (and the material part of calcTSM, fcVM.py:820-1079, when
    disp/sig/pgp are given).  Returns stm,row,col (lower-triangular COO), glv, modf, V,
    loadsums, ne, nn, x."""
    elNodes = np.ascontiguousarray(elNodes, np.int64)
    nocoord = np.ascontiguousarray(nocoord, np.float64)
    ne, nn = len(elNodes), len(nocoord)
    isfix, fixval = fix_arrays(fix, 3 * nn)
    glv = external_loads(nocoord, loads, disp)
    row = np.zeros(465 * ne, np.int64)
    col = np.zeros(465 * ne, np.int64)
    stm = np.zeros(465 * ne, np.float64)
    modf = np.zeros(3 * nn, np.float64)
    x = np.zeros((4 * ne, 3), np.float64)
    V = C.c_double(0.0)
    pgp8 = None if pgp is None else np.ascontiguousarray(pgp, np.uint8)
    pos = lib().orc_calc_stiffness(ne, elNodes, nocoord, _ptr(disp), material[0], material[1],
                                   material[2], isfix, fixval, grav[0], grav[1], grav[2],
                                   _ptr(sig), _ptr(pgp8), Et_E, row, col, stm, glv, modf, x, C.byref(V))
    g3 = glv.reshape(-1, 3)
    return (stm[:pos], row[:pos], col[:pos], glv, modf, V.value, g3[:, 0].sum(), g3[:, 1].sum(),
            g3[:, 2].sum(), ne, nn, x)


def update_stress_load(elNodes, nocoord, material, sig_yield, disp_new, du, sig, sig_update,
                       sig_test, qin, Et_E, LD, pgp):
    """fcVM.py:2197-2464; pgp is a uint8 array updated in place."""
    lib().orc_update_stress_load(len(elNodes), elNodes, nocoord, material[0], material[1],
                                 sig_yield, disp_new, du, sig, sig_update, sig_test, qin, Et_E,
                                 1 if LD else 0, pgp)


def update_PEEQ_CSR(nelem, material, sig_test, sig_new, sig_yield, ultimate_strain, peeq, csr,
                    triax, pressure, sigmises, ecr, Et_E):
    """fcVM.py:2085-2137"""
    lib().orc_update_peeq_csr(nelem, material[0], material[1], sig_test, sig_new, sig_yield,
                              ultimate_strain, peeq, csr, triax, pressure, sigmises, ecr, Et_E)


def node_element_count(elNodes, nn):
    """noce (fcVM.py:183-185): number of elements connected to a node."""
    return np.bincount(np.asarray(elNodes).reshape(-1) - 1, minlength=nn).astype(np.int16)


def mapStresses(averaged, elNodes, nocoord, sig, peeq, sigvm, csr, noce, sig_yield):
    """fcVM.py:2497-2554"""
    nn = len(nocoord)
    t10s = np.zeros((nn, 6))
    t10peeq, t10csr, t10svm, t10triax = np.zeros(nn), np.zeros(nn), np.zeros(nn), np.zeros(nn)
    lib().orc_map_stresses(1 if averaged else 0, len(elNodes), nn, elNodes, sig, peeq, sigvm, csr,
                           np.ascontiguousarray(noce, np.int16), float(sig_yield), t10s, t10peeq,
                           t10csr, t10svm, t10triax)
    return t10s, t10peeq, t10csr, t10svm, t10triax


def calculate_principal_stress(tet10stress):
    """fcVM.py:2954-2994: eigen-decomposition per node, eigenvalues sorted descending,
    vectors scaled by their eigenvalue."""
    s = np.asarray(tet10stress)
    m = np.empty((len(s), 3, 3))
    m[:, 0, 0], m[:, 1, 1], m[:, 2, 2] = s[:, 0], s[:, 1], s[:, 2]
    m[:, 0, 1] = m[:, 1, 0] = s[:, 3]
    m[:, 0, 2] = m[:, 2, 0] = s[:, 4]
    m[:, 1, 2] = m[:, 2, 1] = s[:, 5]
    w, v = np.linalg.eigh(m)
    w = w[:, ::-1]
    v = v[:, :, ::-1]
    return (w[:, 0].copy(), w[:, 1].copy(), w[:, 2].copy(), w[:, 0:1] * v[:, :, 0],
            w[:, 1:2] * v[:, :, 1], w[:, 2:3] * v[:, :, 2])


def buckling_analysis(elNodes, nocoord, material, fix, grav, loads, k=2):
    """Elastic (linear) buckling pre-analysis of the GNLY mode, fcVM.py:1110-1137 + 1195-1214: elastic
    solution for the full load, its Gauss-point stresses, K / G of calcTSM(nstep = 1) and
    eigsh(K, k=2, M=G, sigma=0.1, which='LM', mode='buckling').  Returns (eigenvalues, eigenvectors)."""
    elNodes = np.ascontiguousarray(elNodes, np.int64)
    nocoord = np.ascontiguousarray(nocoord, np.float64)
    ne, nn = len(elNodes), len(nocoord)
    ndof = 3 * nn
    stm, row, col, glv, modf, *_ = calcGSM(elNodes, nocoord, material, fix, grav, loads)
    fixdof = np.ones(ndof)
    fixdof[list(fix.keys())] = 0.0
    ue = _Direct(stm, row, col, ndof)(fixdof * glv + modf)
    sig_new, sig_test = np.zeros(24 * ne), np.zeros(24 * ne)
    pgp = np.zeros(4 * ne, np.uint8)
    update_stress_load(elNodes, nocoord, material, np.full(4 * ne, 1.0e30), np.zeros(ndof), ue, np.zeros(24 * ne), sig_new,
                       sig_test, np.zeros(ndof), 0.0, False, pgp)  # fcVM.py:1195-1197: elastic stresses of ue
    isfix, _ = fix_arrays(fix, ndof)
    r, c = np.zeros(900 * ne, np.int64), np.zeros(900 * ne, np.int64)
    ks, kg = np.zeros(900 * ne), np.zeros(900 * ne)
    lib().orc_buckling_matrices(ne, elNodes, nocoord, material[0], material[1], isfix, sig_new, r, c, ks, kg)
    K = sp.csc_matrix((ks, (r, c)), shape=(ndof, ndof))
    G = -sp.csc_matrix((kg, (r, c)), shape=(ndof, ndof))
    return spla.eigsh(K, k=k, M=G, sigma=0.1, which="LM", mode="buckling")


class _Direct:
    """Stand-in for sksparse.cholmod.cholesky(gsm) (fcVM.py:1121): direct solve of the SPD
    system whose LOWER triangle is given in COO form."""

    def __init__(self, stm, row, col, ndof):
        low = sp.csc_matrix((stm, (row, col)), shape=(ndof, ndof))
        full = low + sp.tril(low, -1).T
        # symmetric-mode SuperLU (minimum degree on A + A^T, diagonal pivots): same solution, about half the fill of the
        # default column ordering on these SPD matrices
        self.lu = spla.splu(full.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))

    def __call__(self, f):
        return self.lu.solve(f)


def calcDisp(elNodes, nocoord, fixdof, movdof, modf, material, stm, row, col, glv, nstep,
             iterat_max, error_max, relax, scale_re, scale_up, scale_dn, sig_yield_inp,
             disp_output, ultimate_strain, Et_E, target_LF, x, noce, fix, grav, loads, gnl="GNLN",
             log=None, max_passes=1, maxImp=0.0, ev1=1.0, ev2=0.0, pass_targets=None):
    """Restatement of fcVM.py:1083-1635 (calcDisp) for gnl == "GNLN" plus the tangent-update
    part of "GNLY".  The GUI hooks (progress bar, matplotlib `plot`, fcVM.py:1104-1108, 1615)
    are replaced by "stop after `max_passes` passes of nstep steps" (the reference's result
    files with nstep+1 rows are exactly one pass)."""
    say = log if log is not None else (lambda *a: None)
    LD = gnl == "GNLY"
    if LD:
        relax, disp_output, scale_up = 1.0, "total", 1.1
    elNodes = np.ascontiguousarray(elNodes, np.int64)
    nocoord = np.ascontiguousarray(nocoord, np.float64)
    ndof, nelem = len(glv), len(elNodes)
    factor = _Direct(stm, row, col, ndof)
    qnorm = np.linalg.norm(glv)
    if qnorm < 1.0:
        qnorm = 1.0
    f = fixdof * glv + modf
    ue = factor(f)
    disp_el = ue.copy()
    dl0 = 1.0 / nstep
    dl = dl0
    du = dl * ue
    sig_new = np.zeros(24 * nelem)
    sig_old = np.zeros(24 * nelem)
    sig_yield = np.full(4 * nelem, float(sig_yield_inp))
    sig_test = np.zeros(24 * nelem)
    peeq, triax, pressure = np.zeros(4 * nelem), np.zeros(4 * nelem), np.zeros(4 * nelem)
    sigmises, ecr, csr = np.zeros(4 * nelem), np.zeros(4 * nelem), np.zeros(4 * nelem)
    pgp = np.zeros(4 * nelem, np.uint8)
    disp_new, disp_old = np.zeros(ndof), np.zeros(ndof)
    lbd = np.zeros(1)
    rfl = np.zeros(1)
    moving = max(movdof) == 1

    def usl(sy, dn, d_u, s_old, s_new, s_test, q, ld):
        update_stress_load(elNodes, nocoord, material, sy, dn, d_u, s_old, s_new, s_test, q, Et_E, ld, pgp)

    if moving:  # fcVM.py:1169-1177
        qelastic = np.zeros(ndof)
        usl(sig_yield, disp_new, ue, sig_old, sig_new, sig_test, qelastic, LD)
        qelastic *= movdof
        qnorm = np.linalg.norm(qelastic)
        sig_new = np.zeros(24 * nelem)
    step, cnt, fail = -1, True, False
    un, csrplot, crip, pplot, svmplot = [0.0], [0.0], [0], [0.0], [0.0]
    triaxplot, peeqplot, peeqmax, ecrplot = [0.0], [0.0], [0.0], [0.0]
    usl(1.0e6 * sig_yield, np.zeros(ndof), ue, sig_old, sig_new, sig_test, np.zeros(ndof), False)  # :1195
    eigenval, eigenvec = np.zeros(1), np.zeros((1, 1))
    maxImp_f = float(maxImp) if LD else 0.0
    if LD and not (float(nstep) > 1.0 and maxImp_f == 0.0):  # fcVM.py:1199-1214
        eigenval, eigenvec = buckling_analysis(elNodes, nocoord, material, fix, grav, loads)
        say("buckling load factors: {}".format(eigenval))
    if float(nstep) == 1.0:  # fcVM.py:1216-1223
        step = 0
        disp_new = ue
        lbd = np.append(lbd, 1.0)
        rfl = np.append(rfl, 1.0)
        un.append(np.max(np.abs(disp_new)))
        cnt = False
    elif LD and maxImp_f != 0.0:  # fcVM.py:1224-1296: imperfection from the two buckling shapes, restart
        e1, e2 = float(ev1), float(ev2)
        ua = e1 / (e1 + e2) * eigenvec[:, 0] + e2 / (e1 + e2) * eigenvec[:, 1]
        ub = e1 / (e1 + e2) * eigenvec[:, 0] - e2 / (e1 + e2) * eigenvec[:, 1]
        ma, mb = np.max(np.abs(ua)), np.max(np.abs(ub))
        if ma > mb:
            imper = maxImp_f / ma * np.sign(ua[np.argmax(np.abs(ua))]) * ua
        else:
            imper = maxImp_f / mb * np.sign(ub[np.argmax(np.abs(ub))]) * ub
        nocoord = nocoord + imper.reshape(-1, 3)
        stm, row, col, glv, modf, *_ = calcGSM(elNodes, nocoord, material, fix, grav, loads)
        qnorm = np.linalg.norm(glv)
        if qnorm < 1.0:
            qnorm = 1.0
        factor = _Direct(stm, row, col, ndof)
        ue = factor(fixdof * glv + modf)
        disp_el = ue.copy()
        dl = dl0
        du = dl * ue
        sig_old = np.zeros(24 * nelem)
        sig_test = np.zeros(24 * nelem)
        disp_new, disp_old = np.zeros(ndof), np.zeros(ndof)
        lbd = np.zeros(1)
    iterat_tot, mrr = 0, False
    sig_new = np.zeros(24 * nelem)
    pgp[:] = 0
    passes = 0
    lout = [0.0]  # fcVM.py:1193: what an nstep == 1 run returns (one all-zero row in the .out table)

    def record():
        dn = disp_new.reshape(-1, 3)
        nd2 = (dn[:(ndof - 1) // 3] ** 2).sum(1)  # fcVM.py:1539-1541: the last node is skipped
        un.append(np.sqrt(max(nd2)))
        update_PEEQ_CSR(nelem, material, sig_test, sig_new, sig_yield, ultimate_strain, peeq, csr,
                        triax, pressure, sigmises, ecr, Et_E)
        maxloc = int(np.argmax(csr))
        csrplot.append(np.max(csr)); crip.append(maxloc); pplot.append(pressure[maxloc])
        svmplot.append(sigmises[maxloc]); triaxplot.append(triax[maxloc]); ecrplot.append(ecr[maxloc])
        peeqplot.append(peeq[maxloc]); peeqmax.append(np.max(peeq))

    while cnt:
        cnt = False
        iRiks = True
        pstep = 0
        while pstep < nstep and not mrr:
            step += 1
            pstep += 1
            restart = 0
            say("Step: {}".format(step))
            a = du.copy()
            if iRiks:
                sig_old = sig_new.copy()
                lbd = np.append(lbd, lbd[step] + dl)
            else:
                lbd[step + 1] = lbd[step] + dl
            qin = np.zeros(ndof)
            usl(sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, LD)
            r = fixdof * lbd[step + 1] * glv - fixdof * qin
            error = np.linalg.norm(r) / qnorm
            iterat = 0
            say("Iteration: {}, Error: {:.2e}".format(iterat, error))
            while error > error_max and not mrr:
                iterat += 1
                iterat_tot += 1
                if LD and (iterat == 1 or np.any(pgp)):  # fcVM.py:1351-1396
                    stm_t, row_t, col_t, glv, modf, *_ = calcGSM(elNodes, nocoord, material, fix, grav,
                                                                 loads, disp=disp_new, sig=sig_old,
                                                                 pgp=pgp, Et_E=Et_E)
                    factor = _Direct(stm_t, row_t, col_t, ndof)
                    ue = factor(fixdof * glv + modf)
                    a = ue.copy()
                    a *= np.linalg.norm(du) / np.linalg.norm(a)
                due = factor(relax * r)
                if iRiks:
                    dl = -np.dot(a, due) / np.dot(a, ue)
                    lbd[step + 1] += dl
                    aa = np.linalg.norm(a)
                else:
                    dl = 0.0
                du += due + dl * ue
                uu = np.linalg.norm(du)
                sf = min(aa / uu, 1.0)
                lbd[step + 1] = lbd[step] + sf * (lbd[step + 1] - lbd[step])
                du *= sf
                qin = np.zeros(ndof)
                usl(sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, LD)
                r = fixdof * (lbd[step + 1] * glv - qin)
                error = np.linalg.norm(r) / qnorm
                say("Iteration: {}, Error: {:.2e}".format(iterat, error))
                if iterat > iterat_max:  # fcVM.py:1457-1484
                    say("RESTART # {}".format(restart + 1))
                    if restart > 3:
                        say("MAXIMUM RESTARTS REACHED")
                        fail = False
                        step -= 1
                        lbd = lbd[:-1]
                        mrr = True
                    restart += 1
                    if step > 0 and not mrr:
                        dl = (lbd[step] - lbd[step - 1]) / scale_re / restart
                        du = (disp_new - disp_old) / scale_re / restart
                    elif not mrr:
                        dl = dl0 / scale_re / restart
                        du = dl * ue / scale_re / restart
                    lbd[step + 1] = lbd[step] + dl
                    if not mrr:
                        qin = np.zeros(ndof)
                        usl(sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, LD)
                        r = fixdof * (lbd[step + 1] * (glv + modf) - qin)
                        error = np.linalg.norm(r) / qnorm
                        iterat = 0
            if abs(target_LF - lbd[step]) < abs(lbd[step + 1] - lbd[step]) and iRiks:  # :1486-1510
                say("REACHED TARGET LOAD")
                fac = (target_LF - lbd[step]) / (lbd[step + 1] - lbd[step])
                du = fac * du
                sig_new = sig_old + fac * (sig_new - sig_old)
                sig_test = sig_old + fac * (sig_test - sig_old)
                lbd[step + 1] = target_LF
                disp_new += du
                record()
                break
            elif not mrr:  # fcVM.py:1515-1559
                disp_old = disp_new.copy()
                disp_new += du
                dl = lbd[step + 1] - lbd[step]
                if moving:
                    rfl = np.append(rfl, np.sum(movdof * qin))
                if iterat > 10:
                    dl /= scale_dn
                    du /= scale_dn
                if iterat < 5:
                    dl *= scale_up
                    du *= scale_up
                record()
                if not iRiks:
                    break
        lout = rfl if moving else lbd
        passes += 1
        cnt = False  # headless stand-in for plot(): stop (fcVM.py:1615)
        if passes < max_passes and not mrr:
            cnt = True
            # "add" button of the plot window, with the target load factor typed into it beforehand (fcVM.py:1664-1672,
            # 1686): pass_targets[i] = the value entered after pass i + 1 (None: field left alone)
            new_t = target_LF
            if pass_targets is not None and passes - 1 < len(pass_targets) and pass_targets[passes - 1] is not None:
                new_t = float(pass_targets[passes - 1])
            LF = lbd[-1]
            if (target_LF - LF) * (new_t - LF) <= 0.0:
                dl = np.sign(new_t - LF) * 1.0 / nstep
                du = dl * ue
            target_LF = new_t
    res = disp_new if disp_output == "total" else disp_new - disp_old
    return dict(displacements=res, disp_el=disp_el, stresses=sig_new, peeq=peeq, sigmises=sigmises,
                csr=csr, lout=np.asarray(lout), un=np.asarray(un), crip=np.asarray(crip),
                peeqplot=np.asarray(peeqplot), pplot=np.asarray(pplot), svmplot=np.asarray(svmplot),
                triaxplot=np.asarray(triaxplot), ecrplot=np.asarray(ecrplot),
                csrplot=np.asarray(csrplot), fail=fail, iterat_tot=iterat_tot, disp_new=disp_new,
                disp_old=disp_old, sig_yield=sig_yield, pgp=pgp, lbd=lbd, eigenval=eigenval, eigenvec=eigenvec,
                nocoord=nocoord)


def read_inp_lines(lines):
    """The 21-line control file as read by fcVM.FCMacro:70-96."""
    t = [s.strip() for s in lines]
    return dict(sig_yield=float(t[0]), grav_x=float(t[1]), grav_y=float(t[2]), grav_z=float(t[3]),
                nstep=int(t[4]), iterat_max=int(t[5]), error_max=float(t[6]), relax=float(t[7]),
                scale_re=float(t[8]), scale_up=float(t[9]), scale_dn=float(t[10]), disp_output=t[11],
                ultimate_strain=float(t[12]), Et_E=float(t[13]), target_LF=float(t[14]),
                csr_option=t[15], averaged_option=t[16], gnl=t[17], maxImp=t[18], ev1=t[19], ev2=t[20])


def run_analysis(elNodes, nocoord, material, fix, loads, ctl, log=None, use_imperfection=False, max_passes=1, pass_targets=None):
    """fcVM.FCMacro:118-175: calcGSM -> calcDisp -> mapStresses, no FreeCAD objects.
    use_imperfection: run the GNLY buckling pre-analysis and apply ctl["maxImp"] (fcVM.py:1199-1296);
    default False = the given coordinates are used as they are."""
    elNodes = np.ascontiguousarray(elNodes, np.int64)
    nocoord = np.ascontiguousarray(nocoord, np.float64)
    nn = len(nocoord)
    fixdof = np.ones(3 * nn)
    movdof = np.zeros(3 * nn)
    for d, v in fix.items():  # fcVM.py:222-258
        fixdof[d] = 0
        if v != 0.0:
            movdof[d] = 1
    grav = (ctl["grav_x"], ctl["grav_y"], ctl["grav_z"])
    stm, row, col, glv, modf, V, lsx, lsy, lsz, ne, nn_, x = calcGSM(elNodes, nocoord, material, fix, grav, loads)
    noce = node_element_count(elNodes, nn)
    res = calcDisp(elNodes, nocoord, fixdof, movdof, modf, material, stm, row, col, glv, ctl["nstep"],
                   ctl["iterat_max"], ctl["error_max"], ctl["relax"], ctl["scale_re"], ctl["scale_up"],
                   ctl["scale_dn"], ctl["sig_yield"], ctl["disp_output"], ctl["ultimate_strain"],
                   ctl["Et_E"], ctl["target_LF"], x, noce, fix, grav, loads, gnl=ctl["gnl"], log=log, max_passes=max_passes, pass_targets=pass_targets,
                   maxImp=ctl.get("maxImp", 0.0) if use_imperfection else 0.0, ev1=ctl.get("ev1", 1.0), ev2=ctl.get("ev2", 0.0))
    nocoord = res["nocoord"]
    t10 = mapStresses(False, elNodes, nocoord, res["stresses"], res["peeq"], res["sigmises"],
                      res["csr"], noce, ctl["sig_yield"])
    res.update(tet10stress=t10[0], tet10peeq=t10[1], tet10csr=t10[2], tet10svm=t10[3],
               tet10triax=t10[4], x=x, V=V, loadsum=(lsx, lsy, lsz), glv=glv, noce=noce)
    return res
