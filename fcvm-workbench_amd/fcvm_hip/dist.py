"""Multi-GPU layer: element range partition, shared-node exchange and a distributed backend with
the same method surface as device.Context, so that solver.RiksDriver (the calcDisp mirror) runs
unchanged on N ranks.

Design (one process per GPU, torch.distributed: "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests):
  * elements are range-partitioned; a rank owns its elements and keeps every node they touch;
  * vectors are kept CONSISTENT: every rank holds the full (assembled) value at each of its local
    nodes.  Element kernels (SpMV with the locally assembled matrix, internal forces, gravity,
    diagonal blocks) produce partial sums at nodes shared with other ranks; `interface_sum`
    exchanges exactly those entries with the neighbouring ranks (point-to-point, no collective on
    the data path) and adds them;
  * inner products weight every dof by 1 on its owner rank (lowest rank that has the node) and 0
    elsewhere, followed by one all-reduce of a scalar.

The local compute is delegated to a `LocalOps` object.  The product implementation is
HipLocalOps (libfcvm_hip kernels on raw device pointers, no fallback); the CPU tests inject an
oracle-backed implementation to exercise partitioning, exchange and the distributed PCG.
"""
import numpy as np
import torch
import torch.distributed as dist


class TVec:
    """A torch-backed distributed vector with the small interface RiksDriver uses."""

    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t

    @property
    def n(self):
        return self.t.numel()

    @property
    def ptr(self):
        return self.t.data_ptr()

    def get(self):
        return self.t.detach().cpu().numpy().copy()

    def clone(self):
        return TVec(self.t.clone())

    def zero(self):
        self.t.zero_()
        return self

    def copy_from(self, o):
        self.t.copy_(o.t)
        return self

    def set(self, host):
        self.t.copy_(torch.as_tensor(np.ascontiguousarray(host)).to(self.t.dtype).reshape(-1))
        return self


class Partition:
    """Node bookkeeping of one rank's element range (built collectively)."""

    def __init__(self, elNodes_global, group=None):
        """elNodes_global: this rank's elements, [ne_loc,10] GLOBAL 1-based node numbers."""
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        el = np.ascontiguousarray(elNodes_global, dtype=np.int64)
        self.gnodes = np.unique(el.reshape(-1))  # sorted global (1-based) ids of the local nodes
        self.nn = len(self.gnodes)
        self.elNodes = (np.searchsorted(self.gnodes, el) + 1).astype(np.int64)  # local 1-based
        lists = [None] * self.world
        dist.all_gather_object(lists, self.gnodes, group=group)
        self.shared = {}  # neighbour rank -> local node indices (0-based), same order on both sides
        owned = np.ones(self.nn, dtype=bool)
        for s in range(self.world):
            if s == self.rank:
                continue
            common = np.intersect1d(self.gnodes, lists[s], assume_unique=True)
            if len(common):
                idx = np.searchsorted(self.gnodes, common)
                self.shared[s] = idx
                if s < self.rank:
                    owned[idx] = False
        self.owned = owned
        self.n_owned = int(owned.sum())
        self.nn_global = int(self._allreduce_int(self.n_owned))

    def _allreduce_int(self, v):
        t = torch.tensor([int(v)], dtype=torch.int64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def local_dofs_of(self, global_dofs):
        """global dof numbers (0-based) -> (mask of those present here, local dof numbers)."""
        g = np.asarray(global_dofs, dtype=np.int64)
        node = g // 3 + 1
        pos = np.searchsorted(self.gnodes, node)
        pos = np.minimum(pos, self.nn - 1)
        here = self.gnodes[pos] == node
        return here, pos[here] * 3 + g[here] % 3


class Exchanger:
    """Adds the partial sums at shared nodes across ranks (ncomp values per node)."""

    def __init__(self, part, device):
        self.part, self.device = part, device
        self.idx = {s: torch.as_tensor(ix, dtype=torch.int64, device=device) for s, ix in part.shared.items()}
        self.bytes_per_call = {}

    def interface_sum(self, t, ncomp):
        """t: flat tensor [nn*ncomp]; in place."""
        p = self.part
        if p.world == 1 or not self.idx:
            return t
        v = t.view(p.nn, ncomp)
        # gloo has no GPU point-to-point: stage through the host there (CPU tests / shared-GPU rehearsal);
        # with nccl (= RCCL) the buffers go GPU to GPU over xGMI
        stage = t.is_cuda and dist.get_backend(p.group) == "gloo"
        sends, recvs, ops = {}, {}, []
        for s, ix in self.idx.items():
            b = v.index_select(0, ix).contiguous()
            sends[s] = b.cpu() if stage else b
            recvs[s] = torch.empty_like(sends[s])
        for s in sorted(self.idx):
            ops.append(dist.P2POp(dist.isend, sends[s], s, group=p.group))
            ops.append(dist.P2POp(dist.irecv, recvs[s], s, group=p.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for s, ix in self.idx.items():
            v.index_add_(0, ix, recvs[s].to(t.device) if stage else recvs[s])
        return t


class DistBackend:
    """Same method surface as device.Context for the Newton/Riks driver, on torch tensors."""

    def __init__(self, part, ops, device, pcg_check_every=25):
        self.part, self.ops, self.device = part, ops, device
        self.ex = Exchanger(part, device)
        self.nn, self.ne = part.nn, len(part.elNodes)
        w = np.repeat(part.owned.astype(np.float64), 3)
        self.w = torch.as_tensor(w, device=device)
        self.group = part.group
        self.check_every = pcg_check_every
        self._timing = dict(spmv_ms=0.0, total_ms=0.0)
        self.dinv_ready = False

    # -- allocation ---------------------------------------------------------------------------
    def _dt(self, dtype):
        return torch.uint8 if np.dtype(dtype) == np.uint8 else torch.float64

    def zeros(self, n, dtype=np.float64):
        return TVec(torch.zeros(int(n), dtype=self._dt(dtype), device=self.device))

    def empty(self, n, dtype=np.float64):
        return TVec(torch.empty(int(n), dtype=self._dt(dtype), device=self.device))

    def array(self, host, dtype=np.float64):
        a = np.ascontiguousarray(host).reshape(-1)
        return TVec(torch.as_tensor(a).to(self._dt(dtype)).to(self.device))

    def sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    # -- reductions ---------------------------------------------------------------------------
    def _allreduce(self, t, op=dist.ReduceOp.SUM):
        if self.part.world > 1:
            dist.all_reduce(t, op=op, group=self.group)
        return t

    def dot_t(self, x, y):
        return self._allreduce(torch.sum(self.w * x * y).reshape(1))

    def dot(self, x, y):
        return float(self.dot_t(x.t, y.t).item())

    def norm(self, x):
        return float(np.sqrt(self.dot(x, x)))

    def loadsum(self, glv):
        v = (self.w * glv.t).view(-1, 3).sum(0)
        return tuple(float(a) for a in self._allreduce(v).cpu())

    def max_node_norm(self, u, nnodes):
        """sqrt(max |u_node|^2) over the global nodes 1..nnodes_global-1 (fcVM.py:1539-1542 skips
        the globally last node); `nnodes` is ignored here, the global count is used."""
        v = u.t.view(-1, 3)
        n2 = (v * v).sum(1)
        last = self.part.nn_global
        keep = torch.as_tensor(self.part.gnodes != last, device=self.device)
        m = torch.max(torch.where(keep, n2, torch.zeros_like(n2))).reshape(1)
        self._allreduce(m, dist.ReduceOp.MAX)
        return float(torch.sqrt(m).item())

    # -- vector algebra (torch plumbing) ------------------------------------------------------
    def axpby(self, a, x, b, y, out=None):
        r = a * x.t if (b == 0.0 or y is None) else a * x.t + b * y.t
        if out is None:
            return TVec(r)
        out.t.copy_(r)
        return out

    def mul(self, x, y, out=None):
        if out is None:
            return TVec(x.t * y.t)
        torch.mul(x.t, y.t, out=out.t)
        return out

    # -- element kernels ----------------------------------------------------------------------
    def assemble(self, grav=(0.0, 0.0, 0.0), disp=None, sig=None, pgp=None, Et_E=0.0, xgp=None):
        glv, modf = self.empty(3 * self.nn), self.empty(3 * self.nn)
        vol = self.ops.assemble(grav, disp, sig, pgp, Et_E, glv, modf, xgp)
        self.ex.interface_sum(glv.t, 3)
        self.ex.interface_sum(modf.t, 3)
        dblk = self.empty(9 * self.nn)
        self.ops.get_diag_blocks(dblk)
        self.ex.interface_sum(dblk.t, 9)
        self.ops.set_diag_blocks(dblk)
        v = torch.tensor([vol], dtype=torch.float64, device=self.device)
        return glv, modf, float(self._allreduce(v).item())

    def update_stress_load(self, sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, Et_E, LD, pgp):
        self.ops.update_stress_load(sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, Et_E, LD, pgp)
        self.ex.interface_sum(qin.t, 3)

    def update_peeq_csr(self, *args):
        o = self.ops.update_peeq_csr(*args)
        if self.part.world == 1:
            return o
        # global argmax(csr) with numpy's first-occurrence rule on the GLOBAL Gauss point index
        o = dict(o)
        o["maxloc"] = int(o["maxloc"]) + 4 * self.ops.element_offset
        allo = [None] * self.part.world
        dist.all_gather_object(allo, o, group=self.group)
        best = max(allo, key=lambda d: (d["csr_max"], -d["maxloc"]))
        best = dict(best)
        best["peeq_max"] = max(d["peeq_max"] for d in allo)
        return best

    def spmv(self, x, y):
        self.ops.spmv(x, y)
        self.ex.interface_sum(y.t, 3)
        return y

    def precond(self, r, z):
        self.ops.precond(r, z)
        return z

    # -- distributed block-Jacobi PCG ---------------------------------------------------------
    def solve(self, f, u=None, rtol=1e-10, maxit=20000):
        if u is None:
            u = self.empty(f.n)
        x, r = u.t, f.t.clone()
        x.zero_()
        z, q = self.empty(f.n), self.empty(f.n)
        self.precond(TVec(r), z)
        p = TVec(z.t.clone())
        rz = self.dot_t(r, z.t)
        bb = self.dot_t(r, r)
        bb_h = float(bb.item())
        info = dict(iterations=0, rel_residual=0.0, spmv=0)
        if bb_h == 0.0:
            return u, info
        thresh = rtol * rtol * bb_h
        it = 0
        while it < maxit:
            for _ in range(self.check_every):
                self.spmv(p, q)
                alpha = rz / self.dot_t(p.t, q.t)
                x.add_(alpha * p.t)
                r.sub_(alpha * q.t)
                self.precond(TVec(r), z)
                rz_new = self.dot_t(r, z.t)
                p.t.mul_(rz_new / rz).add_(z.t)
                rz = rz_new
                it += 1
            rr = float(self.dot_t(r, r).item())
            if not np.isfinite(rr):
                raise RuntimeError("distributed PCG broke down")
            if rr <= thresh:
                break
        info.update(iterations=it, rel_residual=float(np.sqrt(rr / bb_h)), spmv=it)
        if rr > thresh:
            raise RuntimeError("distributed PCG did not converge in %d iterations" % maxit)
        return u, info

    def solve_timing(self):
        return self._timing


class DistModel:
    """Counterpart of solver.GpuModel for one rank of a partitioned mesh."""

    def __init__(self, part, backend, nocoord_local, material, fix_local, loads_local, external_loads_fn):
        self.ctx = backend
        self.part = part
        self.elNodes, self.nocoord = part.elNodes, np.ascontiguousarray(nocoord_local, dtype=np.float64)
        self.material = tuple(float(v) for v in material)
        self.fix = dict(fix_local)
        self.loads = loads_local
        self._ext = external_loads_fn
        self.ne, self.nn = len(part.elNodes), part.nn
        self.ndof = 3 * self.nn
        fixdof, movdof = np.ones(self.ndof), np.zeros(self.ndof)
        for d, v in self.fix.items():
            fixdof[d] = 0.0
            if v != 0.0:
                movdof[d] = 1.0
        self.fixdof, self.movdof = backend.array(fixdof), backend.array(movdof)
        mv = torch.tensor([movdof.max() if len(movdof) else 0.0], dtype=torch.float64, device=backend.device)
        backend._allreduce(mv, dist.ReduceOp.MAX)
        self.moving = bool(mv.item() == 1.0)
        self.t_pattern = 0.0

    def calc_gsm(self, grav, disp=None, sig=None, pgp=None, Et_E=0.0, want_x=True):
        c = self.ctx
        xgp = c.empty(12 * self.ne) if want_x else None
        ext = self._ext(self.nocoord, self.loads, None if disp is None else disp.get())
        glv, modf, V = c.assemble(grav, disp=disp, sig=sig, pgp=pgp, Et_E=Et_E, xgp=xgp)
        # every rank takes part in the exchange, loaded or not (a rank-local "any load?" test would
        # leave the neighbours waiting)
        e = c.array(ext)
        c.ex.interface_sum(e.t, 3)
        glv.t.add_(e.t)
        return glv, modf, V, xgp


class HipLocalOps:
    """LocalOps on libfcvm_hip: the rank's kernels run on torch's current HIP stream, operating on
    the raw device pointers of the torch tensors.  No fallback: needs the library and a GPU."""

    def __init__(self, part, nocoord_local, material, fix_local, device_index, element_offset=0):
        from .device import Context
        self.c = Context(device_index)
        self.c.set_stream(torch.cuda.current_stream(device_index).cuda_stream)
        self.c.set_mesh(part.elNodes, nocoord_local)
        self.c.set_material(*material)
        self.c.set_fix(fix_local)
        self.element_offset = element_offset

    def _p(self, v):
        from .device import DArr
        if v is None:
            return None
        dt = np.uint8 if v.t.dtype == torch.uint8 else np.float64
        return DArr(self.c, v.n, dt, ptr=v.ptr)

    def assemble(self, grav, disp, sig, pgp, Et_E, glv, modf, xgp):
        import ctypes as C
        from ._lib import call
        g = (C.c_double * 3)(*[float(x) for x in grav])
        vol = C.c_double(0.0)
        call("fcvm_assemble", self.c.h, disp.ptr if disp is not None else None, sig.ptr if sig is not None else None,
             pgp.ptr if pgp is not None else None, float(Et_E), g, glv.ptr, modf.ptr,
             xgp.ptr if xgp is not None else None, C.byref(vol))
        return vol.value

    def get_diag_blocks(self, dblk):
        from ._lib import call
        call("fcvm_get_diag_blocks", self.c.h, dblk.ptr)

    def set_diag_blocks(self, dblk):
        from ._lib import call
        call("fcvm_set_diag_blocks", self.c.h, dblk.ptr)

    def update_stress_load(self, sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, Et_E, LD, pgp):
        self.c.update_stress_load(self._p(sig_yield), self._p(disp_new), self._p(du), self._p(sig_old), self._p(sig_new),
                                  self._p(sig_test), self._p(qin), Et_E, LD, self._p(pgp))

    def update_peeq_csr(self, sig_test, sig_new, sig_yield, ultimate_strain, peeq, csr, triax, pressure, sigmises, ecr, Et_E):
        return self.c.update_peeq_csr(self._p(sig_test), self._p(sig_new), self._p(sig_yield), ultimate_strain,
                                      self._p(peeq), self._p(csr), self._p(triax), self._p(pressure), self._p(sigmises),
                                      self._p(ecr), Et_E)

    def spmv(self, x, y):
        self.c.spmv(self._p(x), self._p(y))

    def precond(self, r, z):
        self.c.precond(self._p(r), self._p(z))
