"""Device context and device-resident arrays over the C-ABI (no torch needed on one GPU)."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import call


class DArr:
    """A device array owned by a Context (f64 unless dtype given)."""

    __slots__ = ("ctx", "ptr", "n", "dtype", "_own")

    def __init__(self, ctx, n, dtype=np.float64, ptr=None):
        self.ctx, self.n, self.dtype = ctx, int(n), np.dtype(dtype)
        if ptr is None:
            self.ptr, self._own = ctx._alloc(self.n * self.dtype.itemsize), True
        else:
            self.ptr, self._own = int(ptr), False

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    def zero(self):
        call("fcvm_memset0", self.ctx.h, self.ptr, self.nbytes)
        return self

    def set(self, host):
        a = np.ascontiguousarray(host, dtype=self.dtype).reshape(-1)
        assert a.size == self.n, (a.size, self.n)
        call("fcvm_h2d", self.ctx.h, self.ptr, a.ctypes.data, self.nbytes)
        return self

    def get(self):
        out = np.empty(self.n, dtype=self.dtype)
        call("fcvm_d2h", self.ctx.h, out.ctypes.data, self.ptr, self.nbytes)
        return out

    def copy_from(self, other):
        assert other.nbytes == self.nbytes
        call("fcvm_d2d", self.ctx.h, self.ptr, other.ptr, self.nbytes)
        return self

    def clone(self):
        return DArr(self.ctx, self.n, self.dtype).copy_from(self)

    def free(self):
        if self._own and self.ptr and self.ctx.h:
            self.ctx._release(self.ptr, self.nbytes)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream + one mesh partition (fcvm_ctx)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        call("fcvm_create", int(device), C.byref(h))
        self.h = h
        self.device = device
        self.ne = self.nn = 0
        self._pool, self._pooled = {}, 0
        self._live = set()      # device buffers handed out to DArr objects and not yet released: freed at close()
        # preconditioner of the linear solve: p-multigrid by default, FCVM_PRECOND=bj for block-Jacobi
        self.set_precond(os.environ.get("FCVM_PRECOND", "mg"))
        self.l2_min_coarse_nodes = int(os.environ.get("FCVM_L2_MIN_COARSE", "50"))

    # device buffers of the driver's temporaries are recycled: all work of a context runs on one stream, so a
    # released buffer can be handed out again at once (stream order protects it); hipMalloc/hipFree (+ the
    # synchronisation hipFree implies) would otherwise cost several milliseconds per Newton iteration
    _POOL_LIMIT = 8 << 30

    def _alloc(self, nbytes):
        nbytes = max(int(nbytes), 8)
        lst = self._pool.get(nbytes)
        if lst:
            self._pooled -= nbytes
            ptr = lst.pop()
        else:
            p = C.c_void_p()
            call("fcvm_malloc", self.h, nbytes, C.byref(p))
            ptr = p.value
        self._live.add(ptr)
        return ptr

    def _release(self, ptr, nbytes):
        nbytes = max(int(nbytes), 8)
        self._live.discard(ptr)
        if self._pooled + nbytes <= self._POOL_LIMIT:
            self._pool.setdefault(nbytes, []).append(ptr)
            self._pooled += nbytes
        else:
            call("fcvm_free", self.h, ptr)

    def trim_pool(self):
        for lst in self._pool.values():
            for ptr in lst:
                call("fcvm_free", self.h, ptr)
        self._pool, self._pooled = {}, 0

    def close(self):
        if self.h:
            try:
                self.trim_pool()
                for ptr in self._live:      # arrays still referenced somewhere: their memory goes with the context
                    call("fcvm_free", self.h, ptr)
                self._live = set()
            except Exception:
                pass
            _lib.load().fcvm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- allocation -------------------------------------------------------------------------
    def zeros(self, n, dtype=np.float64):
        return DArr(self, n, dtype).zero()

    def empty(self, n, dtype=np.float64):
        return DArr(self, n, dtype)

    def array(self, host, dtype=np.float64):
        a = np.asarray(host)
        return DArr(self, a.size, dtype).set(a)

    def sync(self):
        call("fcvm_synchronize", self.h)

    def set_stream(self, stream_handle, external=True):
        """Run this context's kernels on an external hipStream_t (0 = HIP's default stream)."""
        call("fcvm_set_stream", self.h, C.c_void_p(stream_handle), 1 if external else 0)

    # -- model ------------------------------------------------------------------------------
    def set_mesh(self, elNodes, nocoord):
        el = np.ascontiguousarray(elNodes, dtype=np.int64)
        xyz = np.ascontiguousarray(nocoord, dtype=np.float64)
        assert el.ndim == 2 and el.shape[1] == 10 and xyz.ndim == 2 and xyz.shape[1] == 3
        call("fcvm_set_mesh", self.h, len(el), len(xyz), el.ctypes.data, xyz.ctypes.data)
        self.ne, self.nn = len(el), len(xyz)

    def set_coords(self, nocoord):
        xyz = np.ascontiguousarray(nocoord, dtype=np.float64)
        assert xyz.shape == (self.nn, 3)
        call("fcvm_set_coords", self.h, xyz.ctypes.data)

    def set_material(self, E, nu, density):
        call("fcvm_set_material", self.h, float(E), float(nu), float(density))

    def set_fix(self, fix):
        """fix: mapping dof -> prescribed value (the reference's `fix` Dict, fcVM.py:222-254)."""
        dofs = np.fromiter(fix.keys(), dtype=np.int64, count=len(fix))
        vals = np.fromiter(fix.values(), dtype=np.float64, count=len(fix))
        call("fcvm_set_fix", self.h, len(dofs), dofs.ctypes.data, vals.ctypes.data)

    def pattern_info(self):
        out = (C.c_int64 * 8)()
        call("fcvm_pattern_info", self.h, out)
        return dict(nnz_blocks=out[0], padded_slots=out[1], slices=out[2], matrix_bytes=out[3],
                    max_row_blocks=out[4], ndof=out[5], spmv_variant=out[7])

    def check_element_map(self):
        """(entries, mismatches) of the device's element -> block-location map against the host builder's rule"""
        out = (C.c_int64 * 2)()
        call("fcvm_check_device_emap", self.h, out)
        return int(out[0]), int(out[1])

    def mem_info(self):
        """(free, total) device memory in bytes"""
        out = (C.c_int64 * 2)()
        call("fcvm_mem_info", self.h, out)
        return int(out[0]), int(out[1])

    def spmv_info(self):
        out = (C.c_int64 * 8)()
        call("fcvm_spmv_info", self.h, out)
        return dict(window_format=bool(out[0]), waves_per_group=out[0], groups=out[1], max_window=out[2], mean_window=out[3], blocks=out[4],
                    bytes=out[5], lds_bytes=out[6], split=out[7])

    def spmv_sym_info(self):
        """the symmetric form of the window format (the PCG's FP64 SpMV): built / stored entries / blocks not stored / bytes per
        launch / LDS bytes / in use"""
        out = (C.c_int64 * 10)()
        call("fcvm_spmv_sym_info", self.h, out)
        return dict(built=bool(out[0]), blocks=out[1], dropped=out[2], bytes=out[3], lds_bytes=out[4], in_use=bool(out[5]),
                    assembly_writes_it=bool(out[6]), assembled_directly=bool(out[7]), declined_graded=bool(out[8]), grading=int(out[9]))

    def placement_info(self):
        """what the placement tuning of the first assembly measured (fcvm_placement_info)"""
        o = (C.c_double * 6)()
        call("fcvm_placement_info", self.h, o)
        return dict(tuned=bool(o[0]), result_vector_ms=(o[1], o[2]), value_array_ms=(o[3], o[4]), spent_ms=o[5])

    def patch_info(self):
        """the vertex-patch smoother of the fine multigrid level (fcvm_patch_info)"""
        o = (C.c_double * 8)()
        call("fcvm_patch_info", self.h, o)
        return dict(built=bool(o[0]), in_use=bool(o[1]), patches=int(o[2]), largest=int(o[3]), mean=o[4], bytes=int(o[5]), lmax=o[6], refresh_ms=o[7])

    def patch_apply(self, x, out=None):
        """out = S x: the undamped vertex-patch smoother of the last assembly (test hook)"""
        out = self.empty(x.n) if out is None else out
        call("fcvm_debug_patch_apply", self.h, x.ptr, out.ptr)
        return out

    def matrix_coo(self):
        nnz = 9 * self.pattern_info()["nnz_blocks"]
        r, c, v = np.empty(nnz, np.int64), np.empty(nnz, np.int64), np.empty(nnz, np.float64)
        call("fcvm_matrix_to_coo", self.h, r.ctypes.data, c.ctypes.data, v.ctypes.data)
        return r, c, v

    # -- hot path ---------------------------------------------------------------------------
    def assemble(self, grav=(0.0, 0.0, 0.0), disp=None, sig=None, pgp=None, Et_E=0.0, xgp=None):
        """calcGSM / calcTSM stiffness part.  Returns (glv_grav, modf, volume) (device, device, float)."""
        glv = self.empty(3 * self.nn)
        modf = self.empty(3 * self.nn)
        g = (C.c_double * 3)(*[float(v) for v in grav])
        vol = C.c_double(0.0)
        call("fcvm_assemble", self.h, disp.ptr if disp is not None else None,
             sig.ptr if sig is not None else None, pgp.ptr if pgp is not None else None, float(Et_E), g,
             glv.ptr, modf.ptr, xgp.ptr if xgp is not None else None, C.byref(vol))
        if os.environ.get("FCVM_MG_LEVEL2", "1") != "0" and self.precond_info()["l2_aggregates"] == 0:
            self.l2_info = self.build_level2()   # once per mesh; tangent re-assemblies keep it
        return glv, modf, vol.value

    def update_stress_load(self, sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, Et_E, LD, pgp):
        call("fcvm_update_stress_load", self.h, sig_yield.ptr, disp_new.ptr if disp_new is not None else None,
             du.ptr, sig_old.ptr, sig_new.ptr, sig_test.ptr, qin.ptr, float(Et_E), 1 if LD else 0, pgp.ptr)

    def update_peeq_csr(self, sig_test, sig_new, sig_yield, ultimate_strain, peeq, csr, triax, pressure,
                        sigmises, ecr, Et_E):
        out = (C.c_double * 8)()
        call("fcvm_update_peeq_csr", self.h, sig_test.ptr, sig_new.ptr, sig_yield.ptr, float(ultimate_strain),
             peeq.ptr, csr.ptr, triax.ptr, pressure.ptr, sigmises.ptr, ecr.ptr, float(Et_E), out)
        return dict(maxloc=int(out[0]), csr_max=out[1], peeq_max=out[2], pressure=out[3], sigmises=out[4],
                    triax=out[5], ecr=out[6], peeq=out[7])

    def map_stresses(self, averaged, sig, peeq, sigvm, csr, sig_yield):
        nn = self.nn
        o = [self.empty(6 * nn)] + [self.empty(nn) for _ in range(4)]
        call("fcvm_map_stresses", self.h, 1 if averaged else 0, sig.ptr, peeq.ptr, sigvm.ptr, csr.ptr,
             float(sig_yield), *[a.ptr for a in o])
        return o

    def principal_stress(self, t10s):
        n = t10s.n // 6
        ps, pv = self.empty(3 * n), self.empty(9 * n)
        call("fcvm_principal_stress", self.h, n, t10s.ptr, ps.ptr, pv.ptr)
        return ps, pv

    def solve(self, f, u=None, rtol=1e-10, maxit=20000):
        """K u = f (replaces CHOLMOD factor(f), fcVM.py:1130).  Returns (u, info dict)."""
        if u is None:
            u = self.empty(f.n)
        info = (C.c_double * 3)()
        call("fcvm_solve", self.h, f.ptr, u.ptr, float(rtol), int(maxit), info)
        return u, dict(iterations=int(info[0]), rel_residual=info[1], spmv=int(info[2]))

    def spmv2(self, xa, xb, ya, yb):
        call("fcvm_spmv2", self.h, xa.ptr, xb.ptr, ya.ptr, yb.ptr)
        return ya, yb

    def mg_apply2(self, ra, rb, za, zb):
        call("fcvm_mg_apply2", self.h, ra.ptr, rb.ptr, za.ptr, zb.ptr)
        return za, zb

    def solve2(self, fa, fb, ua=None, ub=None, rtol=(1e-10, 1e-10), maxit=20000):
        """K ua = fa and K ub = fb in lockstep, every pass over the matrix shared (fcvm_solve2); each result is bitwise what
        solve() returns for that right-hand side.  Returns (ua, ub, info_a, info_b)."""
        ua = ua if ua is not None else self.empty(fa.n)
        ub = ub if ub is not None else self.empty(fb.n)
        info = (C.c_double * 6)()
        call("fcvm_solve2", self.h, fa.ptr, fb.ptr, ua.ptr, ub.ptr, float(rtol[0]), float(rtol[1]), int(maxit), info)
        return (ua, ub, dict(iterations=int(info[0]), rel_residual=info[1], spmv=int(info[2])),
                dict(iterations=int(info[3]), rel_residual=info[4], spmv=int(info[5])))

    # -- linear-buckling pre-analysis (fcvm_hip/buckling.py) ------------------------------------
    def assemble_buckling(self, sig, sigma):
        call("fcvm_assemble_buckling", self.h, sig.ptr, float(sigma))

    def spmv_geometric(self, x, y):
        call("fcvm_spmv_geometric", self.h, x.ptr, y.ptr)
        return y

    def start_vector(self, k, global_dofs=None):
        """start vector number k of the subspace iteration: a fixed pseudo-random function of the GLOBAL dof number, so that
        a partitioned model starts from the same vectors as the whole one"""
        g = np.arange(3 * self.nn, dtype=np.float64) if global_dofs is None else np.asarray(global_dofs, dtype=np.float64)
        v = np.sin(g * (12.9898 + 0.37 * k) + 78.233 * k) * 43758.5453
        return self.array(v - np.floor(v) - 0.5)

    def any_nonzero(self, a):
        """any(a != 0) of a uint8 device array, reduced on the device (4 bytes come back)"""
        out = C.c_int(0)
        call("fcvm_any_u8", self.h, a.n, a.ptr, C.byref(out))
        return bool(out.value)

    def solve_timing(self):
        t = (C.c_double * 4)()
        call("fcvm_solve_timing", self.h, t)
        return dict(spmv_ms=t[0], total_ms=t[1], spmv_timed=int(t[2]), coarse_spmv=int(t[3]))

    def membench(self, nbytes=3 << 30):
        """streaming read rate (GB/s) of a fresh buffer: what the HBM of this GPU delivers right now"""
        o = (C.c_double * 2)()
        call("fcvm_membench", self.h, int(nbytes), o)
        return o[0]

    def set_precond(self, mode="mg", k0=1, k1=60, alpha0=4.0, alpha1=1000.0):
        """mode "bj": 3x3 block-Jacobi; "mg": p-multigrid V-cycle (takes effect at the next assemble)."""
        call("fcvm_set_precond", self.h, 1 if mode == "mg" else 0, int(k0), int(k1), float(alpha0), float(alpha1))

    def precond_info(self):
        o = (C.c_double * 8)()
        call("fcvm_precond_info", self.h, o)
        d = dict(mode="mg" if o[0] == 1 else "bj", coarse_nodes=int(o[1]), lmax0=o[2], lmax1=o[3],
                 coarse_nnz_blocks=int(o[4]), vcycle_fine_spmv=int(o[5]), l2_aggregates=int(o[6]), l2_unknowns=int(o[7]))
        form, wgt, lo = C.c_int(0), C.c_double(0.0), C.c_int(0)
        call("fcvm_get_mg_cycle", self.h, C.byref(form), C.byref(wgt), C.byref(lo))
        # the form that runs (2 needs the 16-bit copy, i.e. the window format; without it the additive form stands in)
        d["cycle_form"], d["cycle_weight"] = form.value, wgt.value
        d["patch_smoother"] = self.patch_info()["in_use"]
        sp = (C.c_int64 * 4)()
        call("fcvm_mg_sparse_info", self.h, sp)
        d["l2_sparse"], d["l3_super_aggregates"], d["l3_unknowns"] = bool(sp[0]), int(sp[2]), int(sp[3])
        if sp[0]:
            d["level2"] = (" -> %d aggregates x rigid-body modes (sparse, %d blocks of 6 x 6, block-Jacobi Chebyshev) -> %d super-aggregates "
                           "(%d unknowns, dense inverse)" % (d["l2_aggregates"], sp[1], sp[2], sp[3]))
        else:
            d["level2"] = (" -> %d aggregates x rigid-body modes (%d unknowns, dense inverse)" % (d["l2_aggregates"], d["l2_unknowns"])
                           if d["l2_aggregates"] else "")
        return d

    def set_mg_cycle(self, additive=True, weight=1.0):
        """form of the level-0 part of the cycle: True / 1 additive, False / 0 multiplicative with FP64 smoother SpMVs, 2 multiplicative
        V(1,1) whose two smoother SpMVs stream a 16-bit fixed-point copy of the matrix (takes effect at the next assemble)"""
        call("fcvm_set_mg_cycle", self.h, int(additive), float(weight))

    def build_level2(self, target=None, k1s=None, alpha1s=None):
        """Third multigrid level (aggregates of tet4 nodes); call after assemble().  Skipped for
        small coarse levels where the long Chebyshev sweep is cheap anyway.  Built inside the library
        (fcvm_mg_build_level2: bookkeeping in C++, Z^T A1 Z and the dense SPD inverse in HIP kernels)."""
        pi = self.precond_info()
        if pi["mode"] != "mg" or pi["coarse_nodes"] < self.l2_min_coarse_nodes:
            return None
        if target is None and os.environ.get("FCVM_MG_TARGET"):
            target = int(os.environ["FCVM_MG_TARGET"])
        if target is None:
            # nodes per aggregate.  The iteration count follows the aggregate SIZE (57 / 66 / 80 PCG iterations for 176 / 350 /
            # 700 nodes on the 1 M cube), the cost of the dense inverse the aggregate COUNT (n^3 at set-up, n^2 per cycle):
            # 176 nodes while that needs no more than 3000 aggregates (18,000 unknowns: 1.3 GB of FP32 inverse, 0.4 s more
            # set-up at 8 M elements, where it brings 70 -> 60 iterations and 496 -> 408 ms per Newton iteration against the
            # 1000 aggregates of round 1; 4900 aggregates: 54 iterations, 397 ms, 3 s of set-up -- profiles/r02x_*); small
            # meshes (cheap either way) go down to 16
            nc = pi["coarse_nodes"]
            max_aggr = int(os.environ.get("FCVM_MG_MAX_AGGREGATES", "3000"))
            if os.environ.get("FCVM_MG_SPARSE", "1") != "0":
                # beyond max_aggr aggregates of 176 nodes the library keeps the aggregate level sparse (fcvm_l2s.inc).
                # Round 3, with the 16-bit multiplicative cycle and the cheaper dense inverse: as small as the dense level
                # carries -- nc / 2600 nodes (the bins over the bounding box then give at most ~2900 aggregates), not below 64:
                # 1 M plate 176 -> 72 nodes = 1090 -> 1970 aggregates: 46.1 -> 39.2 PCG iterations per solve, 3.39 -> 2.9-3.0 s of
                # solves for 50 ms more set-up (the n2 x n2 FP32 product grows from 0.03 to 0.09 ms per iteration; 56 / 48 nodes
                # tip the level into its sparse form: 3.4 / 3.2 s); Embankment 337k: 71.8 -> 68.7 (gpurun_out/r03an, r03ao)
                target = max(16, int(min(176, max(-(-nc // 2600), min(64.0, nc / 200.0)))))
            else:
                target = max(16, -(-nc // max_aggr), int(min(176, nc / 200.0)))
        if k1s is None and os.environ.get("FCVM_MG_K1S"):
            k1s = int(os.environ["FCVM_MG_K1S"])
        if k1s is None:
            # level-1 Chebyshev steps before / after the aggregate correction: larger aggregates leave more for the
            # smoother to do (1 M cube with 1400-node aggregates: 95 PCG iterations with 2 steps, 79 with 3; 2800: 113 / 87 with 4)
            k1s = 2 if target <= 250 else (3 if target <= 1000 else 4)
            # the multiplicative cycle (form 2) leaves the level-1 solve as the weakest part: one more step pays there
            # (1 M plate: 53.7 -> 46.1 PCG iterations, 3.64 -> 3.30 s of solves; Embankment 337k: 82.1 -> 75.1, gpurun_out/r03g)
            if pi.get("cycle_form") == 2 and k1s == 2:
                k1s = 3
        if alpha1s is None and os.environ.get("FCVM_MG_ALPHA1S"):
            alpha1s = float(os.environ["FCVM_MG_ALPHA1S"])
        if alpha1s is None:
            alpha1s = 4.0 * k1s
        out = (C.c_int64 * 2)()
        call("fcvm_mg_build_level2", self.h, int(target), int(k1s), float(alpha1s), out)
        return dict(aggregates=out[0], unknowns=out[1], target=target, inverse="device")

    def spmv(self, x, y):
        call("fcvm_spmv", self.h, x.ptr, y.ptr)
        return y

    def finish_setup(self):
        """wait for / upload what set_mesh left building in the background (otherwise the first assemble does)"""
        call("fcvm_finish_setup", self.h)

    def spmv_lo(self, x, y):
        """y = K~ x with the 16-bit block-scaled copy of the matrix (the smoother's operator in cycle form 2)"""
        call("fcvm_spmv_lo", self.h, x.ptr, y.ptr)
        return y

    def get_diag_blocks(self, out=None):
        """the 3x3 diagonal blocks of the assembled matrix, [nn][9]"""
        out = out if out is not None else self.empty(3 * self.pattern_info()["ndof"])
        call("fcvm_get_diag_blocks", self.h, out.ptr)
        return out

    def mg_apply(self, r, z):
        """z = M^-1 r with the multigrid preconditioner of solve()"""
        call("fcvm_mg_apply", self.h, r.ptr, z.ptr)
        return z

    def precond(self, r, z):
        call("fcvm_precond", self.h, r.ptr, z.ptr)
        return z

    # -- vector helpers ----------------------------------------------------------------------
    def axpby(self, a, x, b, y, out=None):
        out = out if out is not None else self.empty(x.n)
        call("fcvm_vec_axpby", self.h, x.n, float(a), x.ptr, float(b), y.ptr if y is not None else None, out.ptr)
        return out

    def mul(self, x, y, out=None):
        out = out if out is not None else self.empty(x.n)
        call("fcvm_vec_mul", self.h, x.n, x.ptr, y.ptr, out.ptr)
        return out

    def dot(self, x, y):
        r = C.c_double(0.0)
        call("fcvm_vec_dot", self.h, x.n, x.ptr, y.ptr, C.byref(r))
        return r.value

    def mdot(self, xs, y):
        """[x . y for x in xs] in one pass over y and one host synchronisation (bitwise the values of dot())"""
        out = []
        for k0 in range(0, len(xs), 16):
            part = xs[k0:k0 + 16]
            ptrs = (C.c_void_p * len(part))(*[x.ptr for x in part])
            res = (C.c_double * len(part))()
            call("fcvm_vec_mdot", self.h, y.n, len(part), ptrs, y.ptr, res)
            out += list(res)
        return out

    def norm(self, x):
        return float(np.sqrt(self.dot(x, x)))

    def max_node_norm(self, u, nnodes):
        r = C.c_double(0.0)
        call("fcvm_vec_max_node_norm", self.h, int(nnodes), u.ptr, C.byref(r))
        return r.value
