"""Host-side set-up of the third multigrid level: aggregation of the tet4 (level-1) nodes with
rigid-body modes, A2 = Z^T A1 Z and its dense inverse.  O(coarse) work, done once per assembled
coarse operator; the V-cycle itself runs entirely in libfcvm_hip kernels."""
import ctypes as C
import os

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from . import _lib
from ._lib import call


def _aggregate(coord, target):
    """Uniform bins over the bounding box; returns (aggregate id per node, count)."""
    nc = len(coord)
    lo, hi = coord.min(0), coord.max(0)
    ext = np.maximum(hi - lo, 1e-300)
    live = ext > 1e-9 * ext.max()
    nbins = max(1.0, nc / float(target))
    h = (np.prod(ext[live]) / nbins) ** (1.0 / max(1, live.sum()))
    nb = np.where(live, np.maximum(1, np.round(ext / h)), 1).astype(np.int64)
    ij = np.minimum((np.where(live, (coord - lo) / ext, 0.0) * nb).astype(np.int64), nb - 1)
    key = (ij[:, 0] * nb[1] + ij[:, 1]) * nb[2] + ij[:, 2]
    _, agg = np.unique(key, return_inverse=True)
    return agg.astype(np.int32), int(agg.max()) + 1


def build_level2(ctx, target=250, k1s=2, alpha1s=8.0, rank_tol=1e-8):
    pi = ctx.precond_info()
    if pi["mode"] != "mg" or pi["coarse_nodes"] == 0:
        raise RuntimeError("build_level2: assemble with the multigrid preconditioner first")
    nc, nnz = pi["coarse_nodes"], 9 * pi["coarse_nnz_blocks"]
    coord = np.empty((nc, 3))
    fm = np.empty(nc, np.uint8)
    call("fcvm_mg_coarse_nodes", ctx.h, coord.ctypes.data, fm.ctypes.data)
    row, col, val = np.empty(nnz, np.int64), np.empty(nnz, np.int64), np.empty(nnz)
    call("fcvm_mg_coarse_to_coo", ctx.h, row.ctypes.data, col.ctypes.data, val.ctypes.data)
    A1 = sp.csr_matrix((val, (row, col)), shape=(3 * nc, 3 * nc))
    agg, nagg = _aggregate(coord, target)
    order = np.argsort(agg, kind="stable")
    bounds = np.searchsorted(agg[order], np.arange(nagg + 1))
    free = np.stack([(fm & 1) == 0, (fm & 2) == 0, (fm & 4) == 0], axis=1).astype(np.float64)  # [nc,3]
    Z = np.zeros((nc, 3, 6))
    aoff = np.zeros(nagg + 1, np.int64)
    for g in range(nagg):
        idx = order[bounds[g]:bounds[g + 1]]
        x = coord[idx] - coord[idx].mean(0)
        s = max(np.abs(x).max(), 1e-300)
        x = x / s
        B = np.zeros((len(idx), 3, 6))
        B[:, 0, 0] = B[:, 1, 1] = B[:, 2, 2] = 1.0
        B[:, 1, 3], B[:, 2, 3] = -x[:, 2], x[:, 1]   # rotation about x
        B[:, 0, 4], B[:, 2, 4] = x[:, 2], -x[:, 0]   # rotation about y
        B[:, 0, 5], B[:, 1, 5] = -x[:, 1], x[:, 0]   # rotation about z
        B *= free[idx][:, :, None]
        U, S, _ = np.linalg.svd(B.reshape(-1, 6), full_matrices=False)
        r = int((S > rank_tol * max(S[0], 1e-300)).sum()) if S[0] > 0 else 0
        Z[idx, :, :r] = U[:, :r].reshape(len(idx), 3, r)
        aoff[g + 1] = aoff[g] + r
    n2 = int(aoff[nagg])
    if n2 == 0:
        raise RuntimeError("build_level2: no free coarse dofs")
    # sparse Z: rows 3i+c, cols aoff[agg[i]]+k
    ii = np.repeat(np.arange(nc), 18)
    cc = np.tile(np.repeat(np.arange(3), 6), nc)
    kk = np.tile(np.arange(6), 3 * nc)
    ncol = (aoff[1:] - aoff[:-1])[agg]
    keep = kk < np.repeat(ncol, 18)
    Zs = sp.csr_matrix((Z.reshape(-1)[keep], (3 * ii[keep] + cc[keep], aoff[agg][ii[keep]] + kk[keep])), shape=(3 * nc, n2))
    A2 = (Zs.T @ (A1 @ Zs)).toarray()
    A2 = 0.5 * (A2 + A2.T)
    A2[np.diag_indices(n2)] += 1e-13 * np.trace(A2) / n2
    Zc = np.ascontiguousarray(Z.reshape(nc, 18))
    agg = np.ascontiguousarray(agg, np.int32)
    where = "host"
    if os.environ.get("FCVM_L2_DEVICE_INVERSE", "0") == "1":
        # opt-in: dense SPD inverse on the GPU (rocSOLVER potrf + potri inside the library, 0.5 s faster at 6000
        # unknowns once rocBLAS is resident -- but its first use on a fresh box was measured at 6 minutes of
        # library loading, so the default stays the host LAPACK)
        A2 = np.ascontiguousarray(A2)
        rc = _lib.load().fcvm_mg_set_level2_spd(ctx.h, nagg, n2, agg.ctypes.data, aoff.ctypes.data, Zc.ctypes.data, A2.ctypes.data,
                                                int(k1s), float(alpha1s))
        if rc == 0:
            return dict(aggregates=nagg, unknowns=n2, target=target, inverse="device")
        if rc != 2:
            raise _lib.FcvmError("fcvm_mg_set_level2_spd failed: %s" % _lib.load().fcvm_last_error().decode("utf8", "replace"))
    # LAPACK on the host: Cholesky factor + inverse from the factor
    L, info = sla.lapack.dpotrf(A2, lower=1, overwrite_a=0)
    if info != 0:
        raise RuntimeError("build_level2: coarse matrix not positive definite (dpotrf info %d)" % info)
    Li, info = sla.lapack.dpotri(L, lower=1, overwrite_c=1)
    if info != 0:
        raise RuntimeError("build_level2: dpotri info %d" % info)
    # dpotri fills one triangle (the Fortran-order lower one = the upper one of the C-order view handed over below);
    # the other triangle still holds the zeros dpotrf left there, so the sum with the transpose is the full inverse
    A2inv = Li + Li.T
    A2inv[np.diag_indices(n2)] *= 0.5
    A2inv = np.ascontiguousarray(A2inv)
    call("fcvm_mg_set_level2", ctx.h, nagg, n2, agg.ctypes.data, aoff.ctypes.data, Zc.ctypes.data, A2inv.ctypes.data,
         int(k1s), float(alpha1s))
    return dict(aggregates=nagg, unknowns=n2, target=target, inverse=where)
