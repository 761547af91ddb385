"""Device-resident Newton-Raphson / Riks driver: the GPU counterpart of the reference's calcGSM +
calcDisp (fcVM.py:621-816, 1083-1635).

All vectors (displacements, residual, Gauss-point stresses, hardening state) stay in HBM for the
whole analysis; the host only sees scalars (norms, load factor, the per-step result row).  The
CHOLMOD factorisation / back-substitution of the reference is replaced by the block-Jacobi PCG of
libfcvm_hip (`Context.solve`); the control flow -- arc-length correction, restarts, step scaling,
"target load" interpolation and its quirks -- follows the reference line by line so that the
load-displacement curve matches.
"""
import os
import time

import numpy as np

from .device import Context
from .loads import Loads, external_loads


class GpuModel:
    """Mesh + material + boundary conditions + loads on one GPU."""

    def __init__(self, elNodes, nocoord, material, fix, loads=None, device=0, ctx=None):
        self.elNodes = np.ascontiguousarray(elNodes, dtype=np.int64)
        self.nocoord = np.ascontiguousarray(nocoord, dtype=np.float64)
        self.material = tuple(float(v) for v in np.asarray(material).reshape(-1)[:3])
        self.fix = dict(fix)
        self.loads = loads if loads is not None else Loads()
        self.ctx = ctx if ctx is not None else Context(device)
        t0 = time.perf_counter()
        self.ctx.set_mesh(self.elNodes, self.nocoord)
        self.t_pattern = time.perf_counter() - t0
        self.ctx.set_material(*self.material)
        self.ctx.set_fix(self.fix)
        self.ne, self.nn = len(self.elNodes), len(self.nocoord)
        self.ndof = 3 * self.nn
        fixdof = np.ones(self.ndof)
        movdof = np.zeros(self.ndof)
        for d, v in self.fix.items():  # fcVM.py:222-258
            fixdof[d] = 0.0
            if v != 0.0:
                movdof[d] = 1.0
        self.fixdof_h, self.movdof_h = fixdof, movdof
        self.fixdof = self.ctx.array(fixdof)
        self.movdof = self.ctx.array(movdof)
        self.moving = bool(movdof.max() == 1.0) if len(movdof) else False
        self._ext = None   # device copy of the external load vector (None: not built yet, False: no such loads)
        self._follow = None   # follower pressure (GNLY): (FollowerPressure, device copy of the static loads or False)
        self.ctx.finish_setup()   # (the symmetric window form was building beside the lines above)

    def calc_gsm(self, grav, disp=None, sig=None, pgp=None, Et_E=0.0, want_x=True):
        """calcGSM (fcVM.py:621-816) / calcTSM material part: assembles K on the GPU, returns
        (glv, modf, V, xgp): glv = external + gravity loads (device)."""
        c = self.ctx
        xgp = c.empty(12 * self.ne) if want_x else None
        glv_g, modf, V = c.assemble(grav, disp=disp, sig=sig, pgp=pgp, Et_E=Et_E, xgp=xgp)
        # surface / edge / vertex loads (host, O(surface)): only pressure follows the deformed geometry (fcVM.py:858-883), so
        # the displacements come back to the host only then; otherwise the load vector is built once and kept on the device
        follower = disp is not None and self.loads.has_pressure
        if follower:
            # round 5: the static loads once, the pressure from the displacements of the loaded faces' nodes alone (gathered on the device,
            # integrated on the host, scattered back) -- rounds 1-4 took the whole displacement vector to the host at every tangent
            if self._follow is None:
                from ._lib import call     # noqa: F401
                from .loads import FollowerPressure, split_follower
                static, _p = split_follower(self.loads)
                ext = external_loads(self.nocoord, static, None)
                self._follow = (FollowerPressure(self.nocoord, self.loads), c.array(ext) if np.any(ext != 0.0) else False)
                self._follow_idx = c.array(self._follow[0].nodes.astype(np.int64), dtype=np.int64)
                self._follow_buf = c.empty(3 * len(self._follow[0].nodes))
            from ._lib import call
            fp, static_d = self._follow
            n = len(fp.nodes)
            call("fcvm_pack_nodes", c.h, n, self._follow_idx.ptr, 3, disp.ptr, self._follow_buf.ptr)
            f = fp.loads(self._follow_buf.get())
            self._follow_buf.set(f.reshape(-1))
            call("fcvm_scatter_add_nodes", c.h, n, self._follow_idx.ptr, 3, self._follow_buf.ptr, glv_g.ptr)
            ext_d = static_d
        elif self._ext is None:
            ext = external_loads(self.nocoord, self.loads, None)
            ext_d = self._ext = c.array(ext) if np.any(ext != 0.0) else False
        else:
            ext_d = self._ext
        if ext_d is not False:
            c.axpby(1.0, glv_g, 1.0, ext_d, out=glv_g)
        return glv_g, modf, V, xgp


class RiksDriver:
    """State + the three moves of the reference's incremental-iterative loop (fcVM.py:1304-1559):
    start_step() (predictor), iteration() (corrector: solve, arc-length correction, stress update,
    residual) and restart().  calc_disp() strings them together exactly like calcDisp does;
    bench.py drives iteration() directly to time Newton iterations."""

    def __init__(self, model, ctl, log=None, rtol=1e-10, maxit=50000, stats=None):
        self.say = log if log is not None else (lambda *a: None)
        self.model, self.c = model, model.ctx
        c = self.c
        st = self.st = stats if stats is not None else {}
        for k in ("pcg_iterations", "solves", "stress_updates", "newton_iterations", "spmv_launches"):
            st.setdefault(k, 0)
        st.setdefault("solve_s", 0.0)
        st.setdefault("spmv_ms", 0.0)
        self.rtol, self.maxit = rtol, maxit
        self.proj_depth = int(os.environ.get("FCVM_RHS_PROJECTION", "6"))  # previous solutions kept for the start vector
        self._hist, self._gram = [], None
        self._carry = []            # GNLY: solutions of the previous tangent kept as directions for the next start vector
        self.carry_depth = int(os.environ.get("FCVM_GNLY_CARRY", "2"))
        self.nstep, self.iterat_max, self.error_max = int(ctl["nstep"]), int(ctl["iterat_max"]), float(ctl["error_max"])
        self.relax, self.scale_re, self.scale_up, self.scale_dn = (
            float(ctl[k]) for k in ("relax", "scale_re", "scale_up", "scale_dn"))
        self.disp_output, self.ultimate_strain = ctl["disp_output"], float(ctl["ultimate_strain"])
        self.Et_E, self.target_LF, self.sy_inp = float(ctl["Et_E"]), float(ctl["target_LF"]), float(ctl["sig_yield"])
        self.grav = (float(ctl["grav_x"]), float(ctl["grav_y"]), float(ctl["grav_z"]))
        self.LD = ctl.get("gnl", "GNLN") == "GNLY"
        if self.LD:  # fcVM.py:1087-1094
            self.relax, self.disp_output, self.scale_up = 1.0, "total", 1.1
        ndof, nelem = model.ndof, model.ne
        self.ndof, self.nelem = ndof, nelem
        self.glv, self.modf, self.V, self.xgp = model.calc_gsm(self.grav)
        self.glv_h = self.glv.get()
        if hasattr(c, "loadsum"):  # distributed backend: owner-weighted, all-reduced
            self.loadsum = c.loadsum(self.glv)
        else:
            g3 = self.glv_h.reshape(-1, 3)
            self.loadsum = (g3[:, 0].sum(), g3[:, 1].sum(), g3[:, 2].sum())
        self.qnorm = c.norm(self.glv)  # fcVM.py:1115
        if self.qnorm < 1.0:
            self.qnorm = 1.0
        self.f = c.mul(model.fixdof, self.glv)
        c.axpby(1.0, self.f, 1.0, self.modf, out=self.f)
        self.ue = self.solve(self.f)
        self.disp_el = self.ue.clone()
        self.dl0 = 1.0 / self.nstep
        self.dl = self.dl0
        self.du = c.axpby(self.dl, self.ue, 0.0, None)
        self.sig_new, self.sig_old, self.sig_test = c.zeros(24 * nelem), c.zeros(24 * nelem), c.zeros(24 * nelem)
        self.sig_yield = c.array(np.full(4 * nelem, self.sy_inp))
        self.peeq, self.triax, self.pressure = c.zeros(4 * nelem), c.zeros(4 * nelem), c.zeros(4 * nelem)
        self.sigmises, self.ecr, self.csr = c.zeros(4 * nelem), c.zeros(4 * nelem), c.zeros(4 * nelem)
        self.pgp = c.zeros(4 * nelem, np.uint8)
        self.disp_new, self.disp_old = c.zeros(ndof), c.zeros(ndof)
        self.qin, self.r, self.tmp = c.zeros(ndof), c.zeros(ndof), c.zeros(ndof)
        self.a = None
        self.lbd, self.rfl = [0.0], [0.0]
        if model.moving:  # fcVM.py:1169-1177
            self.usl(self.ue)
            c.mul(model.movdof, self.qin, out=self.tmp)
            self.qnorm = c.norm(self.tmp)
            self.sig_new.zero()
        self.step, self.fail, self.mrr = -1, False, False
        self.iterat, self.iterat_tot, self.restart_count, self.aa = 0, 0, 0, 0.0
        self.un, self.csrplot, self.crip, self.pplot, self.svmplot = [0.0], [0.0], [0], [0.0], [0.0]
        self.triaxplot, self.peeqplot, self.peeqmax, self.ecrplot = [0.0], [0.0], [0.0], [0.0]
        self.elastic_only = float(self.nstep) == 1.0
        # linear buckling pre-analysis and imperfection of the GNLY mode (fcVM.py:1199-1214, 1224-1296)
        self.eigenval, self.eigenvec = np.zeros(1), np.zeros((1, 1))
        self.nocoord_old = np.copy(model.nocoord)
        maxImp = float(ctl.get("maxImp", 0.0)) if self.LD else 0.0
        if self.LD and ctl.get("buckling", True) and not (float(self.nstep) > 1.0 and maxImp == 0.0):
            from . import buckling
            self.eigenval, self.eigenvec, _ = buckling.buckling_factors(model, self.grav, log=self.say)
            self.say("buckling load factors: {}".format(self.eigenval))
            if not self.elastic_only and maxImp != 0.0:
                imper = buckling.imperfection(self.eigenvec, float(ctl["ev1"]), float(ctl["ev2"]), maxImp,
                                              global_extreme=getattr(c, "global_extreme", None))
                model.nocoord = model.nocoord + imper.reshape(-1, 3)
                c.set_coords(model.nocoord)
                model._ext = None   # load integrals follow the geometry
                model._follow = None
            # back to the ordinary stiffness matrix (on the imperfect geometry, if any) and its elastic solution
            self._hist, self._gram = [], None
            self.glv, self.modf, self.V, self.xgp = model.calc_gsm(self.grav)
            self.glv_h = self.glv.get()
            self.qnorm = c.norm(self.glv)
            if self.qnorm < 1.0:
                self.qnorm = 1.0
            c.mul(model.fixdof, self.glv, out=self.f)
            c.axpby(1.0, self.f, 1.0, self.modf, out=self.f)
            self.ue = self.solve(self.f)
            self.disp_el = self.ue.clone()
            c.axpby(self.dl, self.ue, 0.0, None, out=self.du)
        if self.elastic_only:  # fcVM.py:1216-1223
            self.step = 0
            self.disp_new = self.ue.clone()
            self.lbd.append(1.0)
            self.rfl.append(1.0)
            self.un.append(float(np.max(np.abs(self.ue.get()))))
        self.sig_new.zero()  # fcVM.py:1301-1302 (unconditional in the reference)
        self.pgp.zero()

    # -- building blocks ----------------------------------------------------------------------
    def dots(self, xs, y):
        """[x . y for x in xs]: one launch and one host synchronisation where the backend has a fused form"""
        c = self.c
        return c.mdot(xs, y) if hasattr(c, "mdot") else [c.dot(x, y) for x in xs]

    def solve(self, rhs, guess=None):
        """K u = rhs to ||rhs - K u|| <= rtol ||rhs||.  guess: a start vector for a matrix that has no history yet (GNLY: the
        solution of the same right-hand side with the previous tangent).  The Newton loop solves with the same matrix over and over
        (the reference back-substitutes with one factorisation); successive right-hand sides are far from
        independent, so the solve starts from the Galerkin projection of rhs onto the last few solutions
        (x0 = U a with (U^T K U) a = U^T rhs).  The stored products K U_i only choose the coefficients a: the residual
        the PCG iteration (and the convergence test) works on is the TRUE one, rhs - K x0, from one extra SpMV, so the
        accuracy of the returned solution does not depend on how accurately the history was solved."""
        t0 = time.perf_counter()
        c, st = self.c, self.st
        hist = self._hist
        x0, r0 = None, rhs
        nf = n0 = 1.0
        if hist and self.proj_depth > 0:
            x0, r0, nf, n0 = self._project(rhs)
        elif guess is not None and self.proj_depth > 0:
            x0 = guess.clone()
            r0 = c.spmv(x0, c.empty(rhs.n))                 # true residual of the start vector
            c.axpby(1.0, rhs, -1.0, r0, out=r0)
            st["projection_spmv"] = st.get("projection_spmv", 0) + 1
            nf, n0 = c.norm(rhs), c.norm(r0)
            if not np.isfinite(n0) or n0 >= nf:
                x0, r0, n0 = None, rhs, nf
        if x0 is not None and n0 <= self.rtol * nf:
            u, info = x0, dict(iterations=0, rel_residual=n0 / nf, spmv=0)
        else:
            d, info = c.solve(r0, rtol=min(1.0, self.rtol * nf / n0) if x0 is not None else self.rtol, maxit=self.maxit)
            u = d if x0 is None else c.axpby(1.0, x0, 1.0, d, out=d)
        # a solution that came out of the projection alone is a combination of the stored ones: keeping it would
        # make the Gram matrix singular without adding a direction
        if self.proj_depth > 0 and info["iterations"] > 0:
            hist.append((u, rhs.clone()))
            if len(hist) > self.proj_depth:
                hist.pop(0)
            m = len(hist)
            g = np.zeros((m, m))
            old = self._gram
            if old is not None and old.shape[0] >= m - 1 and m > 1:
                keep = old[-(m - 1):, -(m - 1):]
                g[:m - 1, :m - 1] = keep
            col = self.dots([h[0] for h in hist], hist[m - 1][1])      # new row / column: U_i . F_new  (symmetric: U^T K U)
            for i in range(m):
                g[i, m - 1] = g[m - 1, i] = col[i]
            self._gram = g
        st["solve_s"] += time.perf_counter() - t0
        st["pcg_iterations"] += info["iterations"]
        st["solves"] += 1
        if "comm_per_iteration" in info:      # distributed backend: exchanges / all-reduces per PCG iteration of the last solve
            st["comm_per_iteration"] = info["comm_per_iteration"]
        if "attainable" in info:              # distributed PCG: the true residual stayed above the tolerance asked for (what FP64 allows
            #                                   for the system, or restarts that kept improving): the worst such solve of the analysis
            st["solves_above_tolerance"] = st.get("solves_above_tolerance", 0) + 1
            st["worst_attained_residual"] = max(st.get("worst_attained_residual", 0.0), float(info["attainable"]))
        if info["iterations"] == 0 and info["spmv"] == 0:
            return u
        tm = self.c.solve_timing()
        st["spmv_ms"] += tm["spmv_ms"]
        st["spmv_launches"] += tm.get("spmv_timed", info["spmv"])  # launches the HIP-event time covers
        st["spmv_total"] = st.get("spmv_total", 0) + info["spmv"]    # incl. the SpMVs inside V-cycles
        return u

    def _project(self, rhs):
        """Start vector for K x = rhs from the history: x0 = U a with (U^T K U) a = U^T rhs, and its TRUE residual (one SpMV).
        Returns (x0, r0, ||rhs||, ||r0||); (None, rhs, nf, nf) when the projection does not help."""
        c, st = self.c, self.st
        U = [h[0] for h in self._hist]
        bn = self.dots(U + [rhs], rhs)          # projections and ||rhs||^2 in one pass over rhs
        b, nf = np.array(bn[:-1]), float(np.sqrt(bn[-1]))
        # least squares with a cut-off: a start vector that is (nearly) a combination of older ones makes the
        # Gram matrix (nearly) singular; np.linalg.solve would answer with huge coefficients instead of failing
        a = np.linalg.lstsq(self._gram, b, rcond=1e-12)[0]
        if not (np.all(np.isfinite(a)) and np.any(a != 0.0)):
            return None, rhs, nf, nf
        x0 = c.axpby(a[0], U[0], 0.0, None)
        for ai, ui in zip(a[1:], U[1:]):
            c.axpby(1.0, x0, ai, ui, out=x0)
        r0 = c.spmv(x0, c.empty(rhs.n))             # true residual of the start vector
        c.axpby(1.0, rhs, -1.0, r0, out=r0)
        st["projection_spmv"] = st.get("projection_spmv", 0) + 1
        n0 = c.norm(r0)
        if not np.isfinite(n0) or n0 >= nf:         # projection did not help: plain solve
            return None, rhs, nf, nf
        return x0, r0, nf, n0

    def _adopt_carry(self, known, skip=None):
        """History for the first solve with a new tangent: `known` = (u, K u) pairs whose product is at hand, plus the carried
        solutions of the previous tangent with their products formed now (one SpMV each).  skip: a carried vector that `known`
        already holds a copy of (the previous ue is both the start vector of the new ue and the first carried solution: taken twice
        it made the Gram matrix singular up to rounding and cost an SpMV for nothing).  Sets _hist / _gram."""
        c = self.c
        carry, self._carry = [u for u in self._carry if u is not skip], []
        if self.proj_depth <= 0 or not carry:
            return
        hist = [(u, f.clone()) for u, f in known]
        while len(carry) >= 2 and hasattr(c, "spmv2"):      # two products per pass over the matrix
            ua, ub = carry.pop(0), carry.pop(0)
            fa, fb = c.spmv2(ua, ub, c.empty(ua.n), c.empty(ub.n))
            hist += [(ua, fa), (ub, fb)]
            self.st["projection_spmv"] = self.st.get("projection_spmv", 0) + 1
        for u in carry:
            hist.append((u, c.spmv(u, c.empty(u.n))))
            self.st["projection_spmv"] = self.st.get("projection_spmv", 0) + 1
        hist = hist[-self.proj_depth:]
        m = len(hist)
        g = np.zeros((m, m))
        U = [h[0] for h in hist]
        for j in range(m):
            col = self.dots(U, hist[j][1])
            for i in range(m):
                g[i, j] = col[i]
        self._hist, self._gram = hist, 0.5 * (g + g.T)

    def solve_pair(self, rhs_a, guess_a, rhs_b):
        """The two back substitutions of a GNLY Newton iteration with its new tangent matrix (fcVM.py:1351-1401: ue = factor(f),
        due = factor(relax * r)) as ONE lockstep solve: every pass over the matrix serves both systems (Context.solve2).
        K ua = rhs_a starts from guess_a (the previous tangent's answer to nearly the same load; true residual from one SpMV),
        K ub = rhs_b from zero.  Returns (ua, ub); backends without solve2 (distributed) solve one after the other."""
        c, st = self.c, self.st
        if not hasattr(c, "solve2") or self.proj_depth <= 0 or os.environ.get("FCVM_SOLVE2", "1") == "0":
            ua = self.solve(rhs_a, guess=guess_a)
            return ua, self.solve(rhs_b)
        t0 = time.perf_counter()
        x0 = guess_a.clone()
        r0 = c.spmv(x0, c.empty(rhs_a.n))                 # true residual of the start vector
        c.axpby(1.0, rhs_a, -1.0, r0, out=r0)
        st["projection_spmv"] = st.get("projection_spmv", 0) + 1
        nf, n0 = c.norm(rhs_a), c.norm(r0)
        if not np.isfinite(n0) or n0 >= nf:
            x0, r0, n0 = None, rhs_a, nf
        if x0 is not None and n0 <= self.rtol * nf:        # the old ue still solves the new system: only due is left
            self._adopt_carry([(x0, rhs_a)], skip=guess_a)   # ... from the projection onto ue (product known: rhs_a) and the carried solutions
            st["solve_s"] += time.perf_counter() - t0
            st["solves"] += 1
            return x0, self.solve(rhs_b)
        rt_a = min(1.0, self.rtol * nf / n0) if x0 is not None else self.rtol
        # due: from the projection onto the carried solutions of the previous tangent (products formed now), if that helps
        xb0, rb0, rt_b = None, rhs_b, self.rtol
        self._adopt_carry([])
        if self._hist:
            xb0, rb0, nfb, n0b = self._project(rhs_b)
            if xb0 is not None:
                rt_b = min(1.0, self.rtol * nfb / max(n0b, 1e-300))
        da, db, ia, ib = c.solve2(r0, rb0, rtol=(rt_a, rt_b), maxit=self.maxit)
        ua = da if x0 is None else c.axpby(1.0, x0, 1.0, da, out=da)
        ub = db if xb0 is None else c.axpby(1.0, xb0, 1.0, db, out=db)
        # history for the start vectors of later solves with this matrix (none in GNLY once Gauss points are plastic: the
        # next iteration assembles a new tangent)
        self._hist = [(ua, rhs_a.clone()), (ub, rhs_b.clone())]
        g = np.zeros((2, 2))
        for j in range(2):
            col = self.dots([ua, ub], self._hist[j][1])
            g[0, j], g[1, j] = col[0], col[1]
        self._gram = 0.5 * (g + g.T)
        st["solve_s"] += time.perf_counter() - t0
        st["pcg_iterations"] += ia["iterations"] + ib["iterations"]
        st["paired_iterations"] = st.get("paired_iterations", 0) + max(ia["iterations"], ib["iterations"])
        st["solves"] += 2
        tm = c.solve_timing()
        st["spmv_ms"] += tm["spmv_ms"]
        st["spmv_launches"] += tm.get("spmv_timed", 0)
        st["spmv_total"] = st.get("spmv_total", 0) + tm.get("spmv_timed", 0)
        return ua, ub

    def usl(self, d_u):
        """update_stress_load with the current state (fcVM.py:1325, 1442, 1478)"""
        self.c.update_stress_load(self.sig_yield, self.disp_new, d_u, self.sig_old, self.sig_new, self.sig_test,
                                  self.qin, self.Et_E, self.LD, self.pgp)
        self.st["stress_updates"] += 1

    def residual(self, lam, with_modf=False):
        """r = fixdof * (lam * glv - qin); returns ||r|| / qnorm (fcVM.py:1329-1341, 1446-1448, 1480-1482)"""
        c = self.c
        c.axpby(lam, self.glv, -1.0, self.qin, out=self.tmp)
        if with_modf:
            c.axpby(1.0, self.tmp, lam, self.modf, out=self.tmp)
        c.mul(self.model.fixdof, self.tmp, out=self.r)
        return c.norm(self.r) / self.qnorm

    def record(self):
        c = self.c
        self.un.append(c.max_node_norm(self.disp_new, (self.ndof - 1) // 3))  # fcVM.py:1539-1542: last node skipped
        o = c.update_peeq_csr(self.sig_test, self.sig_new, self.sig_yield, self.ultimate_strain, self.peeq, self.csr,
                              self.triax, self.pressure, self.sigmises, self.ecr, self.Et_E)
        self.csrplot.append(o["csr_max"]); self.crip.append(o["maxloc"]); self.pplot.append(o["pressure"])
        self.svmplot.append(o["sigmises"]); self.triaxplot.append(o["triax"]); self.ecrplot.append(o["ecr"])
        self.peeqplot.append(o["peeq"]); self.peeqmax.append(o["peeq_max"])

    # -- the loop's moves ---------------------------------------------------------------------
    def start_step(self):
        """fcVM.py:1310-1344: predictor with the current (dl, du); returns the out-of-balance error."""
        self.step += 1
        self.restart_count = 0
        self.say("Step: {}".format(self.step))
        self.a = self.du.clone()
        self.sig_old.copy_from(self.sig_new)
        self.lbd.append(self.lbd[self.step] + self.dl)
        self.usl(self.du)
        self.error = self.residual(self.lbd[self.step + 1])
        self.iterat = 0
        self.say("Iteration: {}, Error: {:.2e}".format(self.iterat, self.error))
        return self.error

    def iteration(self):
        """One Newton iteration, fcVM.py:1346-1455 (without the restart logic)."""
        c, step, lbd = self.c, self.step, self.lbd
        self.iterat += 1
        self.iterat_tot += 1
        self.st["newton_iterations"] += 1
        if self.LD and (self.iterat == 1 or c.any_nonzero(self.pgp)):  # fcVM.py:1351-1396
            self.glv, self.modf, _, _ = self.model.calc_gsm(self.grav, disp=self.disp_new, sig=self.sig_old,
                                                            pgp=self.pgp, Et_E=self.Et_E, want_x=False)
            # new tangent matrix: the stored K U products are stale.  The last solutions themselves still point the way (the
            # right-hand side of the next due is close to the last one's): they are kept, without products -- solve_pair forms
            # K_new u for them (one SpMV each) where that pays (FCVM_GNLY_CARRY vectors, default 2; 0: start every due from zero)
            # (only where solve_pair consumes them: backends without solve2 -- the distributed one -- would hold two fine vectors for nothing)
            self._carry = [h[0] for h in self._hist][-self.carry_depth:] if self.carry_depth > 0 and hasattr(c, "solve2") else []
            self._hist, self._gram = [], None
            c.mul(self.model.fixdof, self.glv, out=self.f)
            c.axpby(1.0, self.f, 1.0, self.modf, out=self.f)
            f2 = c.axpby(self.relax, self.r, 0.0, None)
            # ue (start: the previous tangent's answer to nearly the same load) and due in one lockstep solve
            self.ue, due = self.solve_pair(self.f, self.ue, f2)
            self.a = self.ue.clone()
            c.axpby(c.norm(self.du) / c.norm(self.a), self.a, 0.0, None, out=self.a)
        else:
            c.axpby(self.relax, self.r, 0.0, None, out=self.f)
            due = self.solve(self.f)
        a_due, a_ue, a_a = self.dots([due, self.ue, self.a], self.a)
        self.dl = -a_due / a_ue  # fcVM.py:1414-1417 Riks correction
        lbd[step + 1] += self.dl
        self.aa = float(np.sqrt(a_a))
        c.axpby(1.0, self.du, 1.0, due, out=self.du)
        c.axpby(1.0, self.du, self.dl, self.ue, out=self.du)
        uu = c.norm(self.du)
        sf = min(self.aa / uu, 1.0)
        lbd[step + 1] = lbd[step] + sf * (lbd[step + 1] - lbd[step])
        c.axpby(sf, self.du, 0.0, None, out=self.du)
        self.usl(self.du)
        self.error = self.residual(lbd[step + 1])
        self.say("Iteration: {}, Error: {:.2e}".format(self.iterat, self.error))
        return self.error

    def restart(self):
        """fcVM.py:1457-1484 (called when iterat > iterat_max)."""
        c = self.c
        self.say("RESTART # {}".format(self.restart_count + 1))
        if self.restart_count > 3:
            self.say("MAXIMUM RESTARTS REACHED")
            self.fail = False
            self.step -= 1
            self.lbd = self.lbd[:-1]
            self.mrr = True
        self.restart_count += 1
        step, lbd = self.step, self.lbd
        if step > 0 and not self.mrr:
            self.dl = (lbd[step] - lbd[step - 1]) / self.scale_re / self.restart_count
            k = 1.0 / self.scale_re / self.restart_count
            c.axpby(k, self.disp_new, -k, self.disp_old, out=self.du)
        elif not self.mrr:
            self.dl = self.dl0 / self.scale_re / self.restart_count
            c.axpby(self.dl / self.scale_re / self.restart_count, self.ue, 0.0, None, out=self.du)
        lbd[step + 1] = lbd[step] + self.dl
        if not self.mrr:
            self.usl(self.du)
            self.error = self.residual(lbd[step + 1], with_modf=True)
            self.iterat = 0

    def end_step(self):
        """fcVM.py:1486-1559; returns True when the pass must stop (target load reached)."""
        c, step, lbd = self.c, self.step, self.lbd
        if abs(self.target_LF - lbd[step]) < abs(lbd[step + 1] - lbd[step]):  # fcVM.py:1486-1510
            self.say("REACHED TARGET LOAD")
            fac = (self.target_LF - lbd[step]) / (lbd[step + 1] - lbd[step])
            c.axpby(fac, self.du, 0.0, None, out=self.du)
            c.axpby(fac, self.sig_new, 1.0 - fac, self.sig_old, out=self.sig_new)
            c.axpby(fac, self.sig_test, 1.0 - fac, self.sig_old, out=self.sig_test)
            lbd[step + 1] = self.target_LF
            c.axpby(1.0, self.disp_new, 1.0, self.du, out=self.disp_new)
            self.record()
            return True
        if not self.mrr:  # fcVM.py:1515-1559
            self.disp_old.copy_from(self.disp_new)
            c.axpby(1.0, self.disp_new, 1.0, self.du, out=self.disp_new)
            self.dl = lbd[step + 1] - lbd[step]
            if self.model.moving:
                self.rfl.append(c.dot(self.model.movdof, self.qin))
            if self.iterat > 10:
                self.dl /= self.scale_dn
                c.axpby(1.0 / self.scale_dn, self.du, 0.0, None, out=self.du)
            if self.iterat < 5:
                self.dl *= self.scale_up
                c.axpby(self.scale_up, self.du, 0.0, None, out=self.du)
            self.record()
        return False

    def next_pass(self, new_target=None):
        """What the reference's plot window does between two passes when "add" is clicked, optionally after a new target
        load factor was typed in (fcVM.py:1664-1672, 1686): the step restarts from 1 / nstep of the elastic solution
        when the old target lies at or behind the current load level."""
        new_t = self.target_LF if new_target is None else float(new_target)
        LF = self.lbd[-1]
        if (self.target_LF - LF) * (new_t - LF) <= 0.0:
            self.dl = float(np.sign(new_t - LF)) / self.nstep
            self.c.axpby(self.dl, self.ue, 0.0, None, out=self.du)
        self.target_LF = new_t

    def result(self):
        lout = self.rfl if self.model.moving else self.lbd
        if self.elastic_only:
            lout = [0.0]  # fcVM.py:1193: the load history is only filled inside the step loop
        if self.disp_output == "total":
            res = self.disp_new.get()
        else:
            res = self.disp_new.get() - self.disp_old.get()
        return dict(displacements=res, disp_el=self.disp_el.get(), stresses=self.sig_new, peeq=self.peeq,
                    sigmises=self.sigmises, csr=self.csr, lout=np.asarray(lout), un=np.asarray(self.un),
                    crip=np.asarray(self.crip), peeqplot=np.asarray(self.peeqplot), pplot=np.asarray(self.pplot),
                    svmplot=np.asarray(self.svmplot), triaxplot=np.asarray(self.triaxplot),
                    ecrplot=np.asarray(self.ecrplot), csrplot=np.asarray(self.csrplot), fail=self.fail,
                    iterat_tot=self.iterat_tot, lbd=np.asarray(self.lbd), x=self.xgp.get().reshape(-1, 3), V=self.V,
                    loadsum=self.loadsum, glv=self.glv_h, sig_yield=self.sig_yield, stats=self.st, eigenval=self.eigenval,
                    eigenvec=self.eigenvec, nocoord=self.model.nocoord, nocoord_old=self.nocoord_old)


def calc_disp(model, ctl, log=None, rtol=1e-10, maxit=50000, max_passes=1, stats=None, pass_targets=None):
    """GPU calcDisp (fcVM.py:1083-1635).  ctl: dict as read from the .inp control file
    (inpout.read_inp).  The interactive plot() between passes (fcVM.py:1615) is replaced by
    "continue ('add') until max_passes passes of nstep steps are done"; pass_targets[i] is the target load factor
    typed into the window after pass i + 1 (None / missing: left alone)."""
    d = RiksDriver(model, ctl, log=log, rtol=rtol, maxit=maxit, stats=stats)
    cnt = not d.elastic_only
    passes = 0
    while cnt:
        pstep = 0
        while pstep < d.nstep and not d.mrr:
            pstep += 1
            error = d.start_step()
            while error > d.error_max and not d.mrr:
                error = d.iteration()
                if d.iterat > d.iterat_max:
                    d.restart()
                    error = d.error
            if d.end_step():
                break
        passes += 1
        cnt = passes < max_passes and not d.mrr
        if cnt:
            d.next_pass(pass_targets[passes - 1] if pass_targets is not None and passes - 1 < len(pass_targets) else None)
    return d.result()


class _Stop(Exception):
    pass


def collapse_run(model, ctl, warmup, steps, rtol=1e-10, sync=None):
    """A whole analysis as the reference runs it (fcVM.py:1304-1559: calcGSM, elastic solve, then the arc-length steps of the
    control file with its own error_max -- predictor start_step, Newton iterations, end_step bookkeeping) with a timed
    region cut out of it: Newton iterations warmup+1 .. warmup+steps, with whatever start_step / end_step calls fall between
    them (bench.py's headline).  steps None: no region, the first pass only.  If the control file's steps are used up before
    the region ends, the analysis is continued the way the reference's plot window does ("add": another pass of nstep
    steps).  sync: called before a clock is read (default: the backend's stream synchronisation; the multi-GPU bench adds
    a barrier).  Returns dict(driver, analysis = the whole first pass (what the reference's .out documents), region =
    {s, stats0, stats1} or absent, passes, setup_s)."""
    c = model.ctx
    sync = sync if sync is not None else c.sync
    st = {}
    sync()
    t_begin = time.perf_counter()
    d = RiksDriver(model, ctl, rtol=rtol, stats=st)     # calcGSM + multigrid hierarchy + elastic solve
    sync()
    t_setup = time.perf_counter() - t_begin
    reg = {"n": 0}

    def mark(key):
        sync()
        reg[key] = (time.perf_counter(), dict(st))

    def after_iteration(first_pass_done):
        reg["n"] += 1
        if reg["n"] == warmup:
            mark("start")
        if steps is not None and reg["n"] == warmup + steps:
            mark("end")
            if first_pass_done:
                raise _Stop
    if warmup == 0:
        mark("start")
    analysis, passes = None, 0
    try:
        while True:
            pstep = 0
            while pstep < d.nstep and not d.mrr:
                pstep += 1
                error = d.start_step()
                while error > d.error_max and not d.mrr:
                    error = d.iteration()
                    after_iteration(passes > 0)
                    if d.iterat > d.iterat_max:
                        d.restart()
                        error = d.error
                if d.end_step():
                    break
            passes += 1
            if passes == 1:
                sync()
                dt = time.perf_counter() - t_begin
                analysis = {"analysis_s": dt, "steps": len(d.lbd) - 1, "newton_iterations": st["newton_iterations"],
                            "newton_iterations_per_s": st["newton_iterations"] / dt, "linear_solves": st["solves"],
                            "pcg_iterations_per_solve": st["pcg_iterations"] / max(st["solves"], 1),
                            "collapse_load_factor": float(d.lbd[-1]), "setup_s_inside": t_setup}
            if steps is None or "end" in reg or d.mrr or d.elastic_only:   # steps None: the first pass (the analysis) only
                break
            d.next_pass()
    except _Stop:
        pass
    out = {"driver": d, "analysis": analysis, "passes": passes, "setup_s": t_setup, "stats": st}
    if "start" in reg and "end" in reg:
        (ta, sa), (tb, sb) = reg["start"], reg["end"]
        out["region"] = {"s": tb - ta, "stats0": sa, "stats1": sb}
    return out


def run_analysis(elNodes, nocoord, material, fix, loads, ctl, device=0, log=None, rtol=1e-10, model=None, max_passes=1, pass_targets=None):
    """fcVM.FCMacro:118-175 without FreeCAD: calcGSM -> calcDisp -> mapStresses (all on the GPU).
    Returns the calc_disp dict extended with the nodal (tet10) result fields on the host."""
    if model is None:
        model = GpuModel(elNodes, nocoord, material, fix, loads, device=device)
    res = calc_disp(model, ctl, log=log, rtol=rtol, max_passes=max_passes, pass_targets=pass_targets)
    c = model.ctx
    t10 = c.map_stresses(False, res["stresses"], res["peeq"], res["sigmises"], res["csr"], float(ctl["sig_yield"]))
    ps, pv = c.principal_stress(t10[0])
    res.update(tet10stress=t10[0].get().reshape(-1, 6), tet10peeq=t10[1].get(), tet10csr=t10[2].get(),
               tet10svm=t10[3].get(), tet10triax=t10[4].get(), principal=ps.get().reshape(-1, 3),
               principal_vec=pv.get().reshape(-1, 3, 3), model=model)
    for k in ("stresses", "peeq", "sigmises", "csr", "sig_yield"):
        res[k] = res[k].get()
    return res
