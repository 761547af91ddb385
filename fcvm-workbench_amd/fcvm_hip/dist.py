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
import os

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
    """Adds the partial sums at shared nodes across ranks (ncomp values per node).

    One send buffer holds the values for all neighbours (neighbours in ascending rank order, nodes in ascending global
    number); the received rows are added per interface node in ascending order of the rank they come from, the own
    value standing at the place of the own rank -- every rank that holds a node therefore adds the same numbers in the
    same order and the consistent vectors stay bit-identical on all sharers.  With `ops` offering pack / unpack_sum
    (HipLocalOps: fcvm_pack_nodes / fcvm_unpack_sum) both steps are HIP kernels on the rank's stream; otherwise torch
    (CPU tests)."""

    def __init__(self, part, device, shared=None, nn=None, ops=None):
        self.part, self.device = part, device
        shared = part.shared if shared is None else shared
        self.nn = part.nn if nn is None else nn
        self.ops = ops if (ops is not None and hasattr(ops, "pack")) else None
        self.ranks = sorted(s for s, ix in shared.items() if len(ix))
        self.count = None      # dict with an "exchanges" entry (DistBackend.comm): every call is counted, with or without neighbours
        self._bufs = {}        # (ncomp, dtype, device, staged) -> send buffer, staged send, receive buffer
        self.offs, rows = {}, 0
        for s in self.ranks:
            self.offs[s] = (rows, rows + len(shared[s]))
            rows += len(shared[s])
        self.rows = rows
        # round 5: the exchange as ONE collective -- all_to_all_single with the rows per rank as split sizes (0 for ranks that share nothing): the send
        # buffer is already laid out by ascending neighbour rank, the receive buffer comes back in the same layout.  Over RCCL a batch of point-to-point
        # operations costs ~45 us of host time per exchange (building the P2POps 7, batch_isend_irecv 36), the collective 13
        # (tools/sessions/rccl_p2p_self_r05.py); eight exchange points per PCG iteration are 0.35 ms of host time the one way and 0.1 ms the other,
        # beside 0.2 ms of sequencing and 1.1-1.2 ms of device time per iteration.
        # FCVM_DIST_EXCHANGE=p2p: the batched isend / irecv of rounds 1-4.  A collective has to be entered by EVERY rank of the group, also by one
        # that shares nothing at this level.
        self.alltoall = os.environ.get("FCVM_DIST_EXCHANGE", "alltoall") != "p2p" and part.world > 1
        self.split_rows = [(self.offs[r][1] - self.offs[r][0]) if r in self.offs else 0 for r in range(part.world)]
        if not self.ranks:
            return
        idx_all = np.concatenate([np.asarray(shared[s], dtype=np.int64) for s in self.ranks])
        self.idx_all = torch.as_tensor(idx_all, dtype=torch.int64, device=device)
        # interface nodes and their entry lists (source rank ascending; -1 = the own value)
        inodes = np.unique(idx_all)
        pos = {s: dict(zip(np.asarray(shared[s]).tolist(), range(self.offs[s][0], self.offs[s][1]))) for s in self.ranks}
        eptr, src = [0], []
        for nd in inodes.tolist():
            ent = [(part.rank, -1)] + [(s, pos[s][nd]) for s in self.ranks if nd in pos[s]]
            ent.sort()
            src.extend(e[1] for e in ent)
            eptr.append(len(src))
        self.inodes = torch.as_tensor(inodes, dtype=torch.int64, device=device)
        self.eptr = torch.as_tensor(np.asarray(eptr, dtype=np.int64), device=device)
        self.src = torch.as_tensor(np.asarray(src, dtype=np.int64), device=device)
        cnt = np.diff(eptr)
        self.maxmult = int(cnt.max())
        srcmat = np.full((len(inodes), self.maxmult), -2, dtype=np.int64)     # -2: no entry
        for k in range(self.maxmult):
            m = cnt > k
            srcmat[m, k] = np.asarray(src, dtype=np.int64)[np.asarray(eptr[:-1])[m] + k]
        self.srcmat = torch.as_tensor(srcmat, device=device)

    def begin(self, t, ncomp):
        """First half of interface_sum: pack the shared entries of t and post the sends / receives.  Returns a handle for end(), or None
        when there is nothing to exchange.  Between begin and end the caller may launch work that does not touch the shared entries of t
        (the interior groups of a fine-level pass: fcvm_spmv_dot_part(part = 2)) -- with RCCL the messages then travel on the
        communicator's own stream beside it."""
        p = self.part
        if self.count is not None:
            self.count["exchanges"] += 1
        if p.world == 1 or (not self.ranks and not self.alltoall):
            return None
        # gloo has no GPU point-to-point: stage through the host there (CPU tests / shared-GPU rehearsal);
        # with nccl (= RCCL) the buffers go GPU to GPU over xGMI
        stage = t.is_cuda and dist.get_backend(p.group) == "gloo"
        if self.alltoall and not self.ranks:       # nothing shared at this level: an empty contribution to the collective
            e = torch.empty(0, dtype=t.dtype, device="cpu" if stage else t.device)
            z = [0] * p.world
            return (t, ncomp, stage, None, [dist.all_to_all_single(e, e.clone(), output_split_sizes=z, input_split_sizes=z, group=p.group, async_op=True)])
        key = (ncomp, t.dtype, t.device, stage)
        cached = self._bufs.get(key)
        if cached is None:     # send / receive buffers are allocated once per vector shape and reused: an exchange happens eight times
                               # per PCG iteration and its host-side cost is on the critical path.  (The P2POp list is rebuilt every
                               # time: cached ops keep the process group alive past destroy_process_group and a rank aborted at exit.)
            send = torch.empty(self.rows * ncomp, dtype=t.dtype, device=t.device)
            sbuf = torch.empty(self.rows * ncomp, dtype=t.dtype) if stage else send
            rbuf = torch.empty_like(sbuf)
            cached = self._bufs[key] = (send, sbuf, rbuf)
        send, sbuf, rbuf = cached
        if self.ops is not None and t.is_cuda:
            self.ops.pack(self.idx_all, ncomp, t, send)
        else:
            send.copy_(t.view(self.nn, ncomp).index_select(0, self.idx_all).reshape(-1))
        if stage:
            sbuf.copy_(send)
        if self.alltoall:
            splits = [r * ncomp for r in self.split_rows]
            return (t, ncomp, stage, rbuf, [dist.all_to_all_single(rbuf, sbuf, output_split_sizes=splits, input_split_sizes=splits, group=p.group, async_op=True)])
        ops = []
        for s in self.ranks:
            a, b = self.offs[s]
            ops.append(dist.P2POp(dist.isend, sbuf[a * ncomp:b * ncomp], s, group=p.group))
            ops.append(dist.P2POp(dist.irecv, rbuf[a * ncomp:b * ncomp], s, group=p.group))
        return (t, ncomp, stage, rbuf, dist.batch_isend_irecv(ops))

    def end(self, handle, finish=None):
        """Second half of interface_sum: wait for the messages and add the received rows per interface node in the fixed order.
        finish(recv): a caller's kernel that does the sum itself together with what follows it (DistMG: the smoother step of the level-1
        interface rows, fcvm_unpack_sum_cheb); without neighbours it is called with None."""
        if handle is None:
            if finish is not None:
                finish(None)
            return
        t, ncomp, stage, rbuf, works = handle
        for w in works:
            w.wait()
        if rbuf is None:           # (the empty contribution of a rank that shares nothing at this level)
            if finish is not None:
                finish(None)
            return
        recv = rbuf.to(t.device) if stage else rbuf
        if finish is not None:
            finish(recv)
            return
        if self.ops is not None and t.is_cuda:
            self.ops.unpack_sum(self.inodes, self.eptr, self.src, ncomp, t, recv)
        else:
            v, r = t.view(self.nn, ncomp), recv.view(self.rows, ncomp)
            own = v.index_select(0, self.inodes)
            acc = torch.zeros_like(own)
            for k in range(self.maxmult):       # sequential adds: the same fixed order as the kernel
                col = self.srcmat[:, k]
                term = torch.where((col >= 0).unsqueeze(1), r.index_select(0, col.clamp(min=0)),
                                   torch.where((col == -1).unsqueeze(1), own, torch.zeros_like(own)))
                acc = acc + term
            v.index_copy_(0, self.inodes, acc)

    def interface_sum(self, t, ncomp):
        """t: flat tensor [nn*ncomp]; in place."""
        self.end(self.begin(t, ncomp))
        return t


class DistBackend:
    """Same method surface as device.Context for the Newton/Riks driver, on torch tensors."""

    def __init__(self, part, ops, device, pcg_check_every=25):
        self.part, self.ops, self.device = part, ops, device
        self.ex = Exchanger(part, device, ops=ops)
        # communication steps since the start (shared-node exchanges of any level, all-reduces, device -> host reads of the
        # solver): solve() reports them per PCG iteration in its info, the tests bound them
        self.comm = dict(exchanges=0, allreduces=0, host_syncs=0)
        self.ex.count = self.comm
        self.single_reduction = os.environ.get("FCVM_DIST_CG", "single") != "classic"
        self.nn, self.ne = part.nn, len(part.elNodes)
        w = np.repeat(part.owned.astype(np.float64), 3)
        self.w = torch.as_tensor(w, device=device)
        self.group = part.group
        self.check_every = pcg_check_every
        self._timing = dict(spmv_ms=0.0, total_ms=0.0)
        self._ev_pool, self._ev_list = [], None
        # boundary-groups-first split of the fine-level passes (round 5): the local kernels get the shared nodes once (set_interface_nodes) and
        # launch the groups that hold them first; FCVM_DIST_SPLIT=0: unsplit launches, exchange after the whole pass
        self.split = False
        if hasattr(ops, "set_interface_nodes") and part.world > 1 and os.environ.get("FCVM_DIST_SPLIT", "1") != "0":
            shared_nodes = np.unique(np.concatenate([np.asarray(ix, dtype=np.int64) for ix in part.shared.values()])) if part.shared else np.zeros(0, np.int64)
            self.split_groups = ops.set_interface_nodes(shared_nodes)
            self.split = self.split_groups[0] > 0 and self.split_groups[1] > 0
        self.mg = None  # DistMG once enabled (enable_multigrid)
        self.local_solver = False

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
        self.comm["allreduces"] += 1
        if self.part.world > 1:
            dist.all_reduce(t, op=op, group=self.group)
        return t

    def dot3_t(self, r, z, w):
        """(r.z, w.z, r.r), owner-weighted, ONE all-reduce for the three (single-reduction CG)"""
        out = torch.empty(3, dtype=torch.float64, device=self.device)
        if hasattr(self.ops, "wdot"):
            self.ops.wdot(self.w, r, z, out[0:1])
            self.ops.wdot(self.w, w, z, out[1:2])
            self.ops.wdot(self.w, r, r, out[2:3])
        else:
            wz = self.w * z
            out[0], out[1], out[2] = torch.sum(wz * r), torch.sum(wz * w), torch.sum(self.w * r * r)
        return self._allreduce(out)

    def dot_t(self, x, y):
        """owner-weighted dot product of two consistent vectors, result stays on the device"""
        if hasattr(self.ops, "wdot"):   # one fused pass (fcvm_vec_wdot_dev) instead of two temporaries + a sum
            out = torch.empty(1, dtype=torch.float64, device=self.device)
            self.ops.wdot(self.w, x, y, out)
            return self._allreduce(out)
        return self._allreduce(torch.sum(self.w * x * y).reshape(1))

    def dot(self, x, y):
        return float(self.dot_t(x.t, y.t).item())

    def norm(self, x):
        return float(np.sqrt(self.dot(x, x)))

    # -- linear-buckling pre-analysis (fcvm_hip/buckling.py) ------------------------------------
    def assemble_buckling(self, sig, sigma):
        """K_b + sigma K_g of calcTSM(nstep = 1) on every rank (fcvm_assemble_buckling on the local elements), consistent
        diagonal blocks, multigrid levels refreshed for the shifted operator"""
        self.ops.assemble_buckling(sig, sigma)
        if self.local_solver:
            return
        dblk = self.empty(9 * self.nn)
        self.ops.get_diag_blocks(dblk)
        self.ex.interface_sum(dblk.t, 9)
        self.ops.set_diag_blocks(dblk)
        if self.mg is not None:
            self.mg.setup(buckling=True)

    def spmv_geometric(self, x, y):
        self.ops.spmv_geometric(x, y)
        self.ex.interface_sum(y.t, 3)
        return y

    def start_vector(self, k):
        g = (3 * (self.part.gnodes.astype(np.float64) - 1.0))[:, None] + np.arange(3.0)[None, :]
        v = np.sin(g.reshape(-1) * (12.9898 + 0.37 * k) + 78.233 * k) * 43758.5453
        return self.array(v - np.floor(v) - 0.5)

    def set_coords(self, nocoord_local):
        self.ops.set_coords(nocoord_local)
        if self.mg is not None:   # the library rebuilds its hierarchy for the new geometry: so does the distributed cycle
            self.mg.xyz = np.ascontiguousarray(nocoord_local, dtype=np.float64)
            self.mg.topo, self.mg.nagg, self.mg.ready = False, 0, False

    def global_extreme(self, v):
        """(max |v|, sign of that entry) over all ranks; ties go to the lowest global dof like numpy's argmax"""
        i = int(np.argmax(np.abs(v)))
        mine = (float(np.abs(v[i])), -(3 * int(self.part.gnodes[i // 3] - 1) + i % 3), float(np.sign(v[i])))
        allv = [None] * self.part.world
        dist.all_gather_object(allv, mine, group=self.group)
        best = max(allv)
        return best[0], best[2]

    def any_nonzero(self, a):
        v = (a.t != 0).any().to(torch.float64).reshape(1)
        return bool(self._allreduce(v, dist.ReduceOp.MAX).item() > 0.0)

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
        if self.mg is not None:
            self.mg.setup()
        return glv, modf, float(self._allreduce(v).item())

    def enable_multigrid(self, nocoord_local, **kw):
        """p-multigrid V-cycle as PCG preconditioner on all ranks (needs LocalOps with the mg_* methods).  A single
        rank has nothing to exchange: it runs the library's own V-cycle PCG (fcvm_solve), i.e. exactly the one-GPU
        solver, instead of the host-sequenced cycle."""
        if self.part.world == 1 and hasattr(self.ops, "enable_local_multigrid") and os.environ.get("FCVM_DIST_FORCE_HOST") != "1":
            self.ops.enable_local_multigrid()
            self.local_solver = True
            return
        self.ops.mg_enable()
        self.mg = DistMG(self, nocoord_local, **kw)
        if self.mg.check_every:
            self.check_every = self.mg.check_every

    def update_stress_load(self, sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, Et_E, LD, pgp):
        self.ops.update_stress_load(sig_yield, disp_new, du, sig_old, sig_new, sig_test, qin, Et_E, LD, pgp)
        self.ex.interface_sum(qin.t, 3)

    def update_peeq_csr(self, *args):
        o = self.ops.update_peeq_csr(*args)
        if self.part.world == 1:
            return o
        # global argmax(csr) with numpy's first-occurrence rule on the GLOBAL Gauss point index.  The rank's elements are
        # a range of a PERMUTATION of the caller's list (partition_model's spatial order), so the local Gauss point goes
        # back through element_ids (global number of every local element) -- an offset only describes contiguous ranges
        o = dict(o)
        ml = int(o["maxloc"])
        ids = getattr(self, "element_ids", None)
        if ids is not None:
            if getattr(self, "_ids_ascending", None) is None:
                self._ids_ascending = bool(np.all(np.diff(ids) > 0)) if len(ids) > 1 else True
                if not self._ids_ascending:
                    self._gp_ids = torch.as_tensor(np.repeat(4 * ids, 4) + np.tile(np.arange(4), len(ids)), device=self.device)
            if not self._ids_ascending:
                # ties (csr = 0 everywhere before the first yield, or equal maxima): the kernel's first LOCAL occurrence is
                # not the first GLOBAL one when the local list is a permutation -- take the tied entry with the lowest
                # global Gauss point and re-read the values reported with it
                names = dict(peeq=4, csr_max=5, triax=6, pressure=7, sigmises=8, ecr=9)
                csr_t = args[5].t
                tied = csr_t == csr_t[ml]
                g = int(self._gp_ids[tied].min().item())
                self.comm["host_syncs"] += 1
                if g != 4 * int(ids[ml // 4]) + ml % 4:
                    ml = int(torch.nonzero(self._gp_ids == g)[0].item())
                    for k, a in names.items():
                        if k in o:
                            o[k] = float(args[a].t[ml].item())
            o["maxloc"] = 4 * int(ids[ml // 4]) + ml % 4
        else:
            o["maxloc"] = ml + 4 * self.ops.element_offset
        allo = [None] * self.part.world
        dist.all_gather_object(allo, o, group=self.group)
        best = max(allo, key=lambda d: (d["csr_max"], -d["maxloc"]))
        best = dict(best)
        best["peeq_max"] = max(d["peeq_max"] for d in allo)
        return best

    def spmv(self, x, y, xkx=None):
        """y = K x (consistent);  xkx (1-element device tensor, optional): the rank's share of x^T K x = x_local . (K_local x_local),
        from the SpMV kernel's own inner product -- taken BEFORE the shared-node sum (x consistent, product unsummed: the shares
        add up to x^T K x without owner weights)"""
        evs = self._ev_list
        if evs is not None:
            pool = self._ev_pool
            e0 = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
            e1 = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
            e0.record()
        if xkx is not None and self.split and self.part.world > 1:
            # boundary groups first: every shared entry of the product is complete after part 1, the messages are posted, and the interior
            # groups run while they travel (fcvm_spmv_dot_part; both parts together are bitwise the unsplit launch)
            self.ops.spmv_dot_part(x, y, xkx, 1)
            h = self.ex.begin(y.t, 3)
            self.ops.spmv_dot_part(x, y, xkx, 2)
            if evs is not None:
                e1.record()
                evs.append((e0, e1))
            self.ex.end(h)
            return y
        if xkx is not None:
            self.ops.spmv_dot(x, y, xkx)
        else:
            self.ops.spmv(x, y)
        if evs is not None:
            e1.record()
            evs.append((e0, e1))
        self.ex.interface_sum(y.t, 3)
        return y

    def precond(self, r, z):
        if self.mg is not None and self.mg.ready:
            self.mg.vcycle(r, z)
        else:
            self.ops.precond(r, z)
        return z

    # -- distributed PCG ----------------------------------------------------------------------
    def dot2_t(self, r, z):
        """(r.z, r.r), owner-weighted, one all-reduce for both; stays on the device"""
        out = torch.empty(2, dtype=torch.float64, device=self.device)
        if hasattr(self.ops, "wdot"):
            self.ops.wdot(self.w, r, z, out[0:1])
            self.ops.wdot(self.w, r, r, out[1:2])
        else:
            wr = self.w * r
            out[0], out[1] = torch.sum(wr * z), torch.sum(wr * r)
        return self._allreduce(out)

    def solve(self, f, u=None, rtol=1e-10, maxit=20000):
        """Preconditioned CG on consistent vectors.  Convergence is decided on the device every iteration: once
        r.r <= (rtol ||f||)^2 (in particular r.r == 0) the updates are masked out (alpha = beta = 0), so x and r stay
        what they were at convergence however many iterations the host has already queued; the host reads the flag
        after 1, 1, 2, 4 ... check_every iterations (a Newton step whose start vector is nearly exact converges in one)."""
        if u is None:
            u = self.empty(f.n)
        if self.local_solver:
            info = self.ops.solve_local(f, u, rtol, maxit)
            self._timing = self.ops.c.solve_timing()
            return u, info
        if self.single_reduction:
            try:
                return self._solve_single_reduction(f, u, rtol, maxit)
            finally:
                self._ev_list = None        # (a solve that raised must not leave spmv() recording events)
        x, r = u.t, f.t.clone()
        x.zero_()
        z, q = self.empty(f.n), self.empty(f.n)
        self.precond(TVec(r), z)
        p = TVec(z.t.clone())
        both = self.dot2_t(r, z.t)
        rz, bb = both[0:1].clone(), both[1:2].clone()
        bb_h = float(bb.item())
        info = dict(iterations=0, rel_residual=0.0, spmv=0)
        if bb_h == 0.0:
            return u, info
        if not np.isfinite(bb_h):
            raise RuntimeError("distributed PCG: non-finite right-hand side")
        dev = self.device
        fused = self.mg is not None and self.mg.ready and self.mg.fused
        queued, chunk, spmvs = 0, 1, 0
        finished = False
        if hasattr(self.ops, "pcg_scalars"):
            # scalars on the device in one small array, updated by one single-thread kernel per phase (fcvm_dist_pcg_scalars)
            S = torch.zeros(7, dtype=torch.float64, device=dev)       # rz, rr, thresh, done, iterations, alpha, beta
            S[0:1].copy_(rz); S[1:2].copy_(bb); S[2] = rtol * rtol * bb_h
            alpha, beta = S[5:6], S[6:7]
            while queued < maxit and not finished:
                n = min(chunk, maxit - queued)
                for _ in range(n):
                    self.spmv(p, q)
                    spmvs += 1
                    self.ops.pcg_scalars(0, S, self.dot_t(p.t, q.t))
                    if fused:
                        self.mg.update_and_vcycle(alpha, p, q, TVec(x), TVec(r), z)
                    else:
                        x.addcmul_(p.t, alpha)                  # x += alpha p
                        r.addcmul_(q.t, -alpha)                 # r -= alpha q
                        self.precond(TVec(r), z)
                    self.ops.pcg_scalars(1, S, self.dot2_t(r, z.t))
                    torch.addcmul(z.t, p.t, beta, out=p.t)     # p = z + beta p
                queued += n
                chunk = min(self.check_every, 2 * chunk) if queued >= 2 else 1
                finished = bool(S[3].item() != 0.0)
            Sh = S.cpu()
            rr, it_t, thresh = Sh[1:2], Sh[4:5], Sh[2:3]
        else:
            thresh = torch.full((1,), rtol * rtol * bb_h, dtype=torch.float64, device=dev)
            zero, one = torch.zeros(1, dtype=torch.float64, device=dev), torch.ones(1, dtype=torch.float64, device=dev)
            done = torch.zeros(1, dtype=torch.bool, device=dev)
            it_t = torch.zeros(1, dtype=torch.float64, device=dev)
            rr = bb.clone()
            while queued < maxit and not finished:
                n = min(chunk, maxit - queued)
                for _ in range(n):
                    self.spmv(p, q)
                    spmvs += 1
                    pq = self.dot_t(p.t, q.t)
                    live = ~done
                    alpha = torch.where(live & (pq != 0.0), rz / torch.where(pq != 0.0, pq, one), zero)
                    if fused:
                        self.mg.update_and_vcycle(alpha, p, q, TVec(x), TVec(r), z)
                    else:
                        x.addcmul_(p.t, alpha)                  # x += alpha p
                        r.addcmul_(q.t, -alpha)                 # r -= alpha q
                        self.precond(TVec(r), z)
                    both = self.dot2_t(r, z.t)
                    rz_new, rr_new = both[0:1], both[1:2]
                    beta = torch.where(live & (rz != 0.0), rz_new / torch.where(rz != 0.0, rz, one), zero)
                    torch.addcmul(z.t, p.t, beta, out=p.t)     # p = z + beta p
                    rz = torch.where(live, rz_new, rz)
                    rr = torch.where(live, rr_new, rr)
                    it_t += live
                    done = done | (rr <= thresh) | ~torch.isfinite(rr)
                queued += n
                chunk = min(self.check_every, 2 * chunk) if queued >= 2 else 1
                finished = bool(done.item())
        it = int(it_t.item())
        rr_h = float(rr.item())
        info.update(iterations=it, rel_residual=float(np.sqrt(rr_h / bb_h)) if np.isfinite(rr_h) else float("nan"), spmv=spmvs)
        if not np.isfinite(rr_h):
            raise RuntimeError("distributed PCG broke down")
        if rr_h > float(thresh.item()):
            raise RuntimeError("distributed PCG did not converge in %d iterations" % maxit)
        return u, info

    def _solve_single_reduction(self, f, u, rtol, maxit, max_refine=2):
        """Preconditioned CG in the Chronopoulos-Gear form: the SpMV is applied to z = M^-1 r (w = K z), s = K p follows from
        the recurrence s = w + beta s, and r.z, w.z, r.r of an iteration travel in ONE all-reduce (the classic form needs p.Kp
        before and r.z after the update: two).  Same device-side convergence masking as the classic loop: after
        convergence alpha = beta = 0, queued iterations change nothing.  FCVM_DIST_CG=classic selects the two-reduction loop.
        Both r and s = K p are recurrences here, so the r.r the loop converges on can drift from the true residual at tight
        tolerances: after convergence f - K x is formed once (one more SpMV, exchange and all-reduce); when it misses the
        threshold by more than a factor four in norm the loop restarts from it (at most max_refine = 2 times -- a restart costs a
        third to a half of a solve), and what is reported is the TRUE relative residual.  A second restart that no longer quarters
        the squared residual has found the floor FP64 sets for this system (a tolerance of 1e-12 in some tests lies below it):
        reported as info["attainable"], not an error.  Restarts that are used up while each still improved, with the residual
        above 4 rtol, are reported the same way AND warned about on rank 0; the Newton driver counts both in its statistics
        (solves_above_tolerance, worst_attained_residual).  An iteration budget that runs out raises, on the TRUE residual.
        p.Kp <= 0 or r.z < 0 (an indefinite operator or preconditioner) stop the loop as a breakdown instead of going unnoticed."""
        c0 = dict(self.comm)
        dev = self.device
        self._timing = dict(spmv_ms=0.0, spmv_timed=0, total_ms=0.0)
        x, r = u.t, f.t.clone()
        x.zero_()
        z, w = self.empty(f.n), self.empty(f.n)
        info = dict(iterations=0, rel_residual=0.0, spmv=0, refinements=0)
        fused = self.mg is not None and self.mg.ready and self.mg.fused
        dev_scalars = hasattr(self.ops, "pcg_scalars")
        # round 4: with the fused end of the cycle (DistMG.fused_lo) the rank's shares of r.z and r.r come out of the cycle's last
        # kernel and z^T K z out of the SpMV kernel (its own inner product over the rank's rows, before the shared-node sum): no
        # separate inner-product passes, still ONE all-reduce of three numbers per iteration
        fused_lo = fused and getattr(self.mg, "fused_lo", False)

        # in-solver timing of the rank's fine SpMV kernel (HIP events on the stream it is launched on, around the kernel only: the
        # shared-node exchange that follows is outside), resolved after the solve -- what bench.py's roofline object quotes for N > 1
        timed = dev.type == "cuda"
        evs = self._ev_list = [] if timed else None     # DistBackend.spmv brackets its kernel with events while this list is set

        def spmv_zw():
            if fused_lo:
                self.spmv(z, w, xkx=self.mg._dots[1:2])
            else:
                self.spmv(z, w)

        def dots():
            return self._allreduce(self.mg._dots) if fused_lo else self.dot3_t(r, z.t, w.t)
        total_it, spmvs, queued_all = 0, 0, 0
        bb_h = thresh_h = prev_true = None
        for attempt in range(1 + max_refine):
            self.precond(TVec(r), z)
            spmv_zw()
            spmvs += 1
            g3 = dots()
            gamma, delta = g3[0:1].clone(), g3[1:2].clone()
            if attempt == 0:
                bb_h = float(g3[2].item())
                self.comm["host_syncs"] += 1
                if bb_h == 0.0:
                    info["spmv"] = spmvs
                    return u, info
                if not np.isfinite(bb_h):
                    raise RuntimeError("distributed PCG: non-finite right-hand side")
                thresh_h = rtol * rtol * bb_h
            thresh = torch.full((1,), thresh_h, dtype=torch.float64, device=dev)
            p, sv = TVec(z.t.clone()), TVec(w.t.clone())
            queued, chunk, finished = 0, 1, False
            budget = maxit - queued_all
            if dev_scalars:
                # scalars on the device, one single-thread kernel per iteration (fcvm_dist_pcg_scalars phases 3 / 2) instead of
                # the fifteen one-element tensor operations of the loop below
                S = torch.zeros(7, dtype=torch.float64, device=dev)       # gamma, rr, thresh, done, iterations, alpha, beta
                S[2] = thresh_h
                self.ops.pcg_scalars(3, S, g3)
                alpha, beta = S[5:6], S[6:7]
                one_pass = fused_lo and hasattr(self.ops, "cg_update")
                while queued < budget and not finished:
                    n = min(chunk, budget - queued)
                    for _ in range(n):
                        if one_pass:
                            # the direction updates of the previous iteration ride on this one's x / r update (beta = 0 at the
                            # first: p = z, s = w): one pass over the vectors, then the cycle from its pre-smoothed start
                            self.mg.cg_update_and_vcycle(S, z, w, p, sv, TVec(x), TVec(r))
                            spmv_zw()
                            spmvs += 1
                            self.ops.pcg_scalars(2, S, dots())
                            continue
                        if fused:
                            self.mg.update_and_vcycle(alpha, p, sv, TVec(x), TVec(r), z)
                        else:
                            x.addcmul_(p.t, alpha)
                            r.addcmul_(sv.t, -alpha)
                            self.precond(TVec(r), z)
                        spmv_zw()
                        spmvs += 1
                        self.ops.pcg_scalars(2, S, dots())
                        torch.addcmul(z.t, p.t, beta, out=p.t)       # p = z + beta p
                        torch.addcmul(w.t, sv.t, beta, out=sv.t)     # s = w + beta s   (= K p)
                    queued += n
                    chunk = min(self.check_every, 2 * chunk) if queued >= 2 else 1
                    finished = bool(S[3].item() != 0.0)
                    self.comm["host_syncs"] += 1
                Sh = S.cpu()
                rr_h, it, flag = float(Sh[1]), int(Sh[4]), float(Sh[3])
                detail = " (p.Kp = %.3e, r.M^-1 r = %.3e, r.r / threshold = %.3e)" % (float(Sh[5]), float(Sh[6]), rr_h / thresh_h) if flag == 2.0 else ""
            else:
                zero, one = torch.zeros(1, dtype=torch.float64, device=dev), torch.ones(1, dtype=torch.float64, device=dev)
                done = torch.zeros(1, dtype=torch.bool, device=dev)
                broke = torch.zeros(1, dtype=torch.bool, device=dev)
                it_t = torch.zeros(1, dtype=torch.float64, device=dev)
                rr = g3[2:3].clone()
                broke = broke | (delta <= 0.0) | (gamma < 0.0)
                done = done | broke
                alpha = torch.where(~done, gamma / torch.where(delta != 0.0, delta, one), zero)
                while queued < budget and not finished:
                    n = min(chunk, budget - queued)
                    for _ in range(n):
                        if fused:
                            self.mg.update_and_vcycle(alpha, p, sv, TVec(x), TVec(r), z)   # x += alpha p, r -= alpha s, z = M^-1 r
                        else:
                            x.addcmul_(p.t, alpha)
                            r.addcmul_(sv.t, -alpha)
                            self.precond(TVec(r), z)
                        spmv_zw()
                        spmvs += 1
                        g3 = dots()
                        live = ~done
                        it_t += live
                        rr = torch.where(live, g3[2:3], rr)
                        done = done | (rr <= thresh) | ~torch.isfinite(rr)
                        live = ~done
                        gn = g3[0:1]
                        beta = torch.where(live & (gamma != 0.0), gn / torch.where(gamma != 0.0, gamma, one), zero)
                        den = g3[1:2] - beta * gn / torch.where(alpha != 0.0, alpha, one)
                        bad = live & ((den <= 0.0) | (gn < 0.0))       # p.Kp <= 0 or r.M^-1 r < 0: not SPD
                        broke = broke | bad
                        done = done | bad
                        live = ~done
                        beta = torch.where(live, beta, zero)
                        alpha = torch.where(live & (alpha != 0.0), gn / torch.where(den != 0.0, den, one), zero)
                        gamma = torch.where(live, gn, gamma)
                        torch.addcmul(z.t, p.t, beta, out=p.t)       # p = z + beta p
                        torch.addcmul(w.t, sv.t, beta, out=sv.t)     # s = w + beta s   (= K p)
                    queued += n
                    chunk = min(self.check_every, 2 * chunk) if queued >= 2 else 1
                    finished = bool(done.item())
                    self.comm["host_syncs"] += 1
                rr_h, it, flag = float(rr.item()), int(it_t.item()), 2.0 if bool(broke.item()) else float(done.item())
                detail = ""
            total_it += it
            queued_all += queued
            if not np.isfinite(rr_h):
                raise RuntimeError("distributed PCG broke down (non-finite residual)")
            if flag == 2.0:
                raise RuntimeError("distributed PCG broke down: p.Kp <= 0 or r.M^-1 r < 0 (operator or preconditioner not positive "
                                   "definite) after %d iterations%s" % (total_it, detail))
            # the true residual of what the recurrences converged to (or stopped at, when the iteration budget ran out)
            out_of_budget = rr_h > thresh_h
            self.spmv(TVec(x), w)
            spmvs += 1
            rt = f.t - w.t
            rr_true = float(self.dot_t(rt, rt).item())
            self.comm["host_syncs"] += 1
            rr_h = rr_true
            if out_of_budget or rr_true <= 16.0 * thresh_h or attempt == max_refine or queued_all >= maxit:
                break
            if prev_true is not None and rr_true > 0.25 * prev_true:
                # a restart from the true residual no longer halves it: this is the floor FP64 sets for this matrix and solution
                # (eps |K| |x| against |f|: 6.7e-10 of |f| on the shifted buckling operator of the column test, whatever rtol asks),
                # not a drifted recurrence -- a direct solver's residual would sit there too.  Reported, not an error.
                info["attainable"] = float(np.sqrt(rr_true / bb_h))
                break
            prev_true = rr_true
            r = rt
            info["refinements"] += 1
        info.update(iterations=total_it, rel_residual=float(np.sqrt(rr_h / bb_h)), spmv=spmvs)
        self._ev_list = None
        if timed:       # (the host has just read scalars of this stream: the events are complete)
            torch.cuda.current_stream(dev).synchronize()
            self._timing = dict(spmv_ms=float(sum(a.elapsed_time(b) for a, b in evs)), spmv_timed=len(evs), total_ms=0.0)
            for a, b in evs:
                self._ev_pool += [a, b]
        info["comm"] = {k: self.comm[k] - c0[k] for k in self.comm}
        info["comm_per_iteration"] = {k: (self.comm[k] - c0[k]) / max(queued_all + 1, 1) for k in ("exchanges", "allreduces")}
        if rr_h > 16.0 * thresh_h and "attainable" not in info:
            if queued_all >= maxit:
                raise RuntimeError("distributed PCG did not converge in %d iterations (true relative residual %.3e)"
                                   % (maxit, np.sqrt(rr_h / bb_h)))
            info["attainable"] = float(np.sqrt(rr_h / bb_h))       # (max_refine restarts, each still improving: stop here)
            if self.part.rank == 0:
                import warnings
                warnings.warn("distributed PCG: true relative residual %.3e after %d restarts, asked for %.1e (recurrences drifting, "
                              "not the FP64 floor)" % (info["attainable"], info["refinements"], rtol))
        return u, info

    def solve_timing(self):
        return self._timing


LMAX_SAFETY = 1.02     # on the Lanczos estimate of lambda_max(Dinv A): MG_LMAX_SAFETY of csrc/fcvm_mg.inc, see the comment there


class _Cheb:
    """Chebyshev coefficients for eigenvalues of Dinv*A in [lmax/alpha, lmax] (same recurrences as
    the single-GPU V-cycle in csrc/fcvm_mg.inc)."""

    def __init__(self, lmax, alpha, k):
        lmin = lmax / alpha
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        self.c0 = 1.0 / theta
        self.cdd, self.cdr = [0.0] * (k + 1), [0.0] * (k + 1)
        rho = 1.0 / sigma
        for j in range(1, k + 1):
            rn = 1.0 / (2.0 * sigma - rho)
            self.cdd[j], self.cdr[j] = rn * rho, 2.0 * rn / delta
            rho = rn


class DistMG:
    """The three-level V-cycle of the single-GPU solver (tet10 -> tet4 -> aggregates with rigid-body
    modes) on a partitioned mesh.  Vectors are consistent; every local SpMV is followed by the
    shared-node exchange of its level; restriction weights every node by 1/multiplicity so that the
    partial sums add up.  The aggregate level is replicated: its right-hand side is all-reduced (one small all-reduce
    per cycle) and every rank computes the same correction from the same operator -- a dense inverse up to 3000
    aggregates, the sparse level with super-aggregates beyond (see _level2) -- at the one-GPU aggregate size."""

    def __init__(self, be, nocoord_local, k0=1, alpha0=4.0, k1s=None, alpha1s=None, target=None, check_every=4, additive=None,
                 weight=1.0, max_dense=None):
        self.be, self.ops, self.part = be, be.ops, be.part
        self.xyz = np.ascontiguousarray(nocoord_local, dtype=np.float64)
        # form of the level-0 part, as on one GPU (csrc/fcvm_mg.inc): 2 = multiplicative V(1,1), one damped block-Jacobi step before
        # and after the coarse correction, its two fine products on the 16-bit copy of every rank's matrix and summed over the
        # shared nodes (round 3; default where the local kernels offer spmv_lo); 1 = additive (rounds 1-2: no fine product inside
        # the preconditioner); 0 = multiplicative Chebyshev on the FP64 matrix.  FCVM_DIST_CYCLE=additive forces form 1.
        if additive is None:
            additive = not hasattr(be.ops, "spmv_lo") or os.environ.get("FCVM_DIST_CYCLE", "") == "additive"
            self.form = 1 if additive else 2
        else:
            self.form = 1 if additive else 0
        if k1s is None and os.environ.get("FCVM_DIST_K1S"):
            # round 5: two steps per side are SIX exchange points per PCG iteration instead of eight (each level-1 product is one); what that costs
            # in iterations is in profiles/r05_dist_rehearsal_final_build.txt, what an exchange costs on xGMI has never been measured
            k1s = max(1, int(os.environ["FCVM_DIST_K1S"]))
        if k1s is None:
            k1s = 3 if self.form == 2 else 2         # the multiplicative cycle leaves the level-1 solve as the weakest part
        if alpha1s is None:
            alpha1s = 4.0 * k1s
        if self.form == 2:
            alpha0 = max(alpha0, 16.0)               # damping of the Jacobi step as in csrc/fcvm_mg.inc (mg_smoother_weight)
        self.k0, self.alpha0, self.k1s, self.alpha1s, self.target = k0, alpha0, k1s, alpha1s, target
        self.max_dense = max_dense                   # most aggregates with a dense inverse (None: FCVM_MG_MAX_AGGREGATES, 3000)
        self.check_every = check_every
        self.additive, self.weight = self.form == 1, weight
        self.fused = self.form in (1, 2)             # the PCG update x, r and the start of the cycle in one kernel
        # round 4 (form 2 on the HIP kernels): the residual after pre-smoothing formed inside the restriction, the end of the cycle
        # (subtraction, Dinv, addition) in one kernel that also leaves the rank's shares of r.z and r.r, z^T K z from the SpMV
        # kernel's own inner product: per PCG iteration eight vector passes and four launches fewer.  FCVM_DIST_FUSED_LO=0: off.
        self.fused_lo = (self.form == 2 and all(hasattr(be.ops, m) for m in ("post_smooth", "mg_restrict_diff", "jacobi6f", "spmv_dot",
                                                                              "pcg_update_xr_z0_f32"))
                         and os.environ.get("FCVM_DIST_FUSED_LO", "1") != "0")
        self.rz_rr = None                            # (r.z, r.r) shares of the last cycle (device, 2 one-element views), fused_lo only
        self.lo_resid = self.fused_lo and hasattr(be.ops, "spmv_lo_resid") and os.environ.get("FCVM_DIST_LO_RESID", "1") != "0"
        # round 4: the last product of the level-1 pre-smoothing only ever enters (a) the residual the aggregate level restricts --
        # whose sum over the ranks the all-reduce of r2 completes anyway -- and (b) the residual the post-smoothing starts from, where
        # it rides on the exchange of the FIRST post-smoothing product (both are subtracted from the same vector): one shared-node
        # exchange less per cycle (8 instead of 9 exchange points per PCG iteration), the same numbers up to rounding.
        # FCVM_DIST_MERGE_L1=0: every product exchanged on its own.
        self.merge_l1 = os.environ.get("FCVM_DIST_MERGE_L1", "1") != "0"
        self.ready = False
        self.topo = False
        self.lmax0 = self.lmax1 = 0.0
        self.lmax_stale = True
        self.timing = {}

    # -- set-up -------------------------------------------------------------------------------
    def _topology(self):
        be, part, dev = self.be, self.part, self.be.device
        c2f = np.asarray(self.ops.mg_coarse_map(), dtype=np.int64)
        self.nc = len(c2f)
        f2c = np.full(part.nn, -1, dtype=np.int64)
        f2c[c2f] = np.arange(self.nc)
        shared1 = {}
        for s, ix in part.shared.items():
            cix = f2c[ix]
            shared1[s] = cix[cix >= 0]  # shared corner nodes, same (global id) order on both sides
        self.ex1 = Exchanger(part, dev, shared=shared1, nn=self.nc, ops=self.ops)
        self.ex1.count = be.comm
        ones = torch.ones(part.nn, dtype=torch.float64, device=dev)
        be.ex.interface_sum(ones, 1)
        self.w0 = (1.0 / ones).repeat_interleave(3)           # 1 / (ranks holding the node), per dof
        c2f_t = torch.as_tensor(c2f, device=dev)
        self.own1 = be.w.view(-1, 3)[c2f_t].reshape(-1).contiguous()  # owner weights of level-1 dofs
        self.c2f = c2f
        n0, n1 = 3 * part.nn, 3 * self.nc
        mk = lambda n: TVec(torch.zeros(n, dtype=torch.float64, device=dev))  # noqa: E731
        self.r0, self.d0, self.t0 = mk(n0), mk(n0), mk(n0)
        self.x2, self.z0 = mk(n0), mk(n0)       # form 2: x after the coarse correction; a scratch vector
        self.r1, self.x1, self.rr1, self.d1, self.t1 = mk(n1), mk(n1), mk(n1), mk(n1), mk(n1)
        self.tk1 = mk(n1)      # the last pre-smoothing product of level 1, unsummed (merge_l1)
        # round 5: level-1 passes on the 16-bit kernel with the smoother step as epilogue (fcvm_mg_l1_pass); the rows of shared corner nodes
        # are deferred to the kernel that sums their exchanged products.  Needs the fused form-2 cycle and the merged level-1 exchange;
        # FCVM_DIST_FUSED_L1=0 keeps the round-4 sequence (FP32 sliced-ELL product + step kernel per pass)
        self.fused_l1 = (self.fused_lo and self.merge_l1 and all(hasattr(self.ops, m) for m in ("mg_l1_pass", "mg_l1_spmv", "unpack_sum_cheb"))
                         and self.ops.mg_l1_16() and os.environ.get("FCVM_DIST_FUSED_L1", "1") != "0")
        if self.fused_l1:
            self.dn1 = mk(n1)
            defer = np.zeros(self.nc, dtype=np.uint8)
            if self.ex1.ranks:
                defer[self.ex1.inodes.cpu().numpy()] = 1
            self.defer1 = torch.as_tensor(defer, device=dev)
            # boundary slices first (round 5): the slices of the 16-bit copy with a row of a shared corner node are launched before the exchange of the
            # raw products is posted, the others beside it; FCVM_DIST_L1_SPLIT=0: one launch, exchange after it
            self.l1_split = False
            if (self.ex1.ranks and hasattr(self.ops, "mg_l1_set_interface") and self.part.world > 1
                    and os.environ.get("FCVM_DIST_L1_SPLIT", "1") != "0"):
                nb, ni = self.ops.mg_l1_set_interface(self.ex1.inodes.cpu().numpy())
                self.l1_split = nb > 0 and ni > 0
                self.l1_split_slices = (nb, ni)
        self.topo = True
        self.lmax_stale = True

    def _lmax(self, level, n, wdot):
        """largest eigenvalue of Dinv*A: Lanczos for the pencil (A, D) with D-orthonormal vectors v_j = Dinv p_j carried beside
        p_j = D v_j (the recurrences of mg_estimate_lmax, csrc/fcvm_mg.inc, with the products summed over the shared nodes and
        owner-weighted dots).  Rounds 1-3 ran 15 power iterations from a smooth function of the global node number: 87 % of the
        eigenvalue on the plate, and with the 1.1 safety factor a Jacobi weight of c0 * lambda_max = 2.02 on two ranks -- cycle
        form 2 was indefinite there (r.M^-1 r < 0 after seven PCG iterations; three ranks sat at 1.97 and took 43 instead of 33
        iterations per solve)."""
        dev = self.be.device
        g = torch.as_tensor(self.part.gnodes if level == 0 else self.part.gnodes[self.c2f], dtype=torch.float64, device=dev)
        # start vector: a fixed pseudo-random function of the GLOBAL node number (the same vector on every partition), rich
        # in short wavelengths whatever the numbering
        ph = torch.remainder(g * 0.6180339887498949 + 0.31, 1.0)
        P = TVec((torch.stack([torch.remainder(ph * 7919.0, 1.0), torch.remainder(ph * 104729.0 + 0.17, 1.0),
                               torch.remainder(ph * 1299709.0 + 0.71, 1.0)], 1) - 0.5).reshape(-1).contiguous())
        V, Q, U, tmp1, tmp2 = (TVec(torch.zeros(n, dtype=torch.float64, device=dev)) for _ in range(5))
        ex = self.be.ex if level == 0 else self.ex1
        dinv = lambda x, y: self.ops.mg_cheb_init(level, x, tmp1, tmp2, y, 1.0)      # noqa: E731  y = Dinv x
        dinv(P, V)
        nrm = float(wdot(P.t, V.t).item())
        P.t.mul_(1.0 / np.sqrt(nrm)); V.t.mul_(1.0 / np.sqrt(nrm))
        kmax = int(os.environ.get("FCVM_DIST_LMAX_ITERS", "40"))
        al, be_, lam, hist, beta_prev = [], [], 0.0, [], 0.0
        for j in range(kmax):
            self.ops.mg_spmv(level, V, U)
            ex.interface_sum(U.t, 3)
            alpha = float(wdot(V.t, U.t).item())
            al.append(alpha)
            T = np.diag(al) + (np.diag(be_, 1) + np.diag(be_, -1) if be_ else 0.0)
            lam = float(np.linalg.eigvalsh(T)[-1])
            hist.append(lam)
            # (at least 16 steps and 3e-4 over two steps: this estimator, unlike mg_estimate_lmax, was never checked against an
            # eigensolver until round 5 -- tests/test_distributed.py does now -- and the bounds are formed once per mesh)
            if (j + 1 >= 16 and abs(lam - hist[-3]) <= 3e-4 * abs(lam)) or j + 1 == kmax:
                break
            Q.t.mul_(-beta_prev).add_(U.t).add_(P.t, alpha=-alpha)      # p~ = u - alpha p_j - beta_{j-1} p_{j-1}
            dinv(Q, U)                                                  # v~
            b2 = float(wdot(Q.t, U.t).item())
            if b2 <= 1e-28 * alpha * alpha:
                break
            beta = float(np.sqrt(b2))
            Q.t.mul_(1.0 / beta); U.t.mul_(1.0 / beta)
            be_.append(beta)
            beta_prev = beta
            V, U = U, V
            P, Q = Q, P
        if not np.isfinite(lam) or lam <= 0.0:
            raise RuntimeError("DistMG: eigenvalue estimate broke down")
        return lam

    def _lap(self, name, t0):
        """stage times of the set-up (FCVM_SETUP_TIMING=1: printed by rank 0; kept in self.timing either way)"""
        import time
        self.be.sync()
        t = time.perf_counter()
        self.timing[name] = self.timing.get(name, 0.0) + t - t0
        return t

    def setup(self, buckling=False):
        """after every assembly: consistent level-1 diagonal blocks; smoother bounds and the aggregate level once per mesh (tangent
        re-assemblies keep both, as the one-GPU solver does -- csrc/fcvm_mg.inc mg_setup: plastic softening only lowers the eigenvalues of
        Dinv A; rounds 1-4 re-ran 2 x 40 Lanczos steps with an exchange and an all-reduce each at every GNLY tangent).  buckling: the
        shifted operator K_b + sigma K_g gets bounds of its own, and so does the ordinary assembly after it."""
        import time
        be, dev = self.be, self.be.device
        t0 = time.perf_counter()
        if not self.topo:
            self._topology()
            t0 = self._lap("level-1 topology, exchange tables", t0)
        # level-1 diagonal blocks: local partial sums -> consistent -> inverted for the smoother
        dblk = TVec(torch.zeros(9 * self.nc, dtype=torch.float64, device=dev))
        self.ops.mg_get_diag(1, dblk)
        self.ex1.interface_sum(dblk.t, 9)
        self.ops.mg_set_diag(1, dblk)
        if buckling or self.lmax_stale or not (self.lmax0 and self.lmax1):
            wdot0 = lambda a, b: be._allreduce(torch.sum(be.w * a * b).reshape(1))  # noqa: E731
            wdot1 = lambda a, b: be._allreduce(torch.sum(self.own1 * a * b).reshape(1))  # noqa: E731
            self.lmax0 = self._lmax(0, 3 * self.part.nn, wdot0)
            self.lmax1 = self._lmax(1, 3 * self.nc, wdot1)
            self.lmax_stale = bool(buckling)
            if os.environ.get("FCVM_VERBOSE") and self.part.rank == 0:
                print("fcvm dist: smoother bounds lmax0 %.6f lmax1 %.6f (alpha0 %.3g, weight %.3g)" % (self.lmax0, self.lmax1, self.alpha0, self.weight), flush=True)
            t0 = self._lap("smoother bounds (2 Lanczos runs)", t0)
        self.ch0 = _Cheb(LMAX_SAFETY * self.lmax0, self.alpha0, self.k0)
        self.ch1 = _Cheb(LMAX_SAFETY * self.lmax1, self.alpha1s, self.k1s)
        if not getattr(self, "nagg", 0):
            self._level2()   # once per mesh; tangent re-assemblies keep the aggregate level (as device.Context.assemble does on one GPU)
            t0 = self._lap("aggregate level", t0)
            if os.environ.get("FCVM_SETUP_TIMING") and self.part.rank == 0:
                print("fcvm dist set-up: " + ", ".join("%s %.3f s" % kv for kv in self.timing.items()), flush=True)
        self.ready = True

    def _level2(self):
        """Aggregate level of the partitioned mesh, REPLICATED on every rank (round 5; rounds 1-4 kept a dense n2 x n2 matrix per rank,
        all-reduced it in 1 GiB slabs and inverted it in full everywhere: 13 GB and 41 x the one-GPU inverse at 8 x 1 M tet10).
        Aggregates = bins over the GLOBAL bounding box at the one-GPU aggregate size (~72 level-1 nodes at 1 M tet10 per rank) for any rank
        count; rigid-body modes about the global aggregate centroids, orthonormalised with Gram matrices summed over the ranks; the rank's
        share of A2 = Z^T A1 Z as 6 x 6 blocks from the device (ops.mg_galerkin_blocks: k_galerkin_blocks on the rank's level-1 matrix);
        the union of the ranks' block patterns and ONE all-reduce of the block values (288 B per block: ~90 MB at 8 x 1 M tet10); then every
        rank builds the same level from the same numbers (ops.mg_l2_finish): a dense inverse while the aggregates number at most
        FCVM_MG_MAX_AGGREGATES (3000), beyond that the sparse level of the one-GPU library (block-Jacobi Chebyshev sweeps on A2 closed by a
        dense level of super-aggregates, csrc/fcvm_l2s.inc) with the super-aggregates' modes summed over the ranks here.  Per cycle: one
        all-reduce of r2 (6 doubles per aggregate), the same correction computed by every rank, no gather."""
        import time
        be, dev, part = self.be, self.be.device, self.part
        t0 = time.perf_counter()
        coord, fm = self.ops.mg_coarse_nodes()
        own = self.own1.view(-1, 3)[:, 0].cpu().numpy() > 0.5
        lo = torch.as_tensor(coord.min(0), device=dev)
        hi = torch.as_tensor(coord.max(0), device=dev)
        be._allreduce(lo, dist.ReduceOp.MIN)
        be._allreduce(hi, dist.ReduceOp.MAX)
        lo, hi = lo.cpu().numpy(), hi.cpu().numpy()
        ncg = torch.tensor([float(own.sum())], dtype=torch.float64, device=dev)
        ncg = float(be._allreduce(ncg).item())
        # nodes per aggregate: the rule of the one-GPU solver (device.Context.build_level2) on the MEAN level-1 node count of a rank, so
        # the aggregate size -- which is what the PCG iteration count follows -- does not grow with the rank count
        ncr = ncg / part.world
        target = self.target if self.target else int(os.environ.get("FCVM_MG_TARGET", "0"))
        if not target:
            target = max(16, int(min(176, max(-(-int(ncr) // 2600), min(64.0, ncr / 200.0)))))
        ext = np.maximum(hi - lo, 1e-300)
        live = ext > 1e-9 * ext.max()
        h = (np.prod(ext[live]) / max(1.0, ncg / target)) ** (1.0 / max(1, live.sum()))
        nb = np.where(live, np.maximum(1, np.round(ext / h)), 1).astype(np.int64)
        ij = np.minimum((np.where(live, (coord - lo) / ext, 0.0) * nb).astype(np.int64), nb - 1)
        key = (ij[:, 0] * nb[1] + ij[:, 1]) * nb[2] + ij[:, 2]
        keys = [None] * part.world
        dist.all_gather_object(keys, np.unique(key), group=part.group)
        allk = np.unique(np.concatenate(keys))
        agg = np.searchsorted(allk, key)
        nagg = len(allk)

        def modes(cen, sc, of):
            """raw rigid-body modes [node][component][mode] about cen[of], lengths scaled by sc[of], supported dofs masked"""
            x = (coord - cen[of]) / np.maximum(sc, 1e-300)[of][:, None]
            R = np.zeros((self.nc, 3, 6))
            R[:, 0, 0] = R[:, 1, 1] = R[:, 2, 2] = 1.0
            R[:, 1, 3], R[:, 2, 3] = -x[:, 2], x[:, 1]
            R[:, 0, 4], R[:, 2, 4] = x[:, 2], -x[:, 0]
            R[:, 0, 5], R[:, 1, 5] = -x[:, 1], x[:, 0]
            free = np.stack([(fm & 1) == 0, (fm & 2) == 0, (fm & 4) == 0], 1).astype(np.float64)
            return R * free[:, :, None]

        def centroids(of, n):
            """global centroid and half extent of every group (owner-weighted sums / maxima over the ranks)"""
            acc = np.zeros((n, 4))
            np.add.at(acc, of[own], np.concatenate([coord[own], np.ones((own.sum(), 1))], 1))
            acc_t = torch.as_tensor(acc, device=dev)
            be._allreduce(acc_t)
            acc = acc_t.cpu().numpy()
            cen = acc[:, :3] / np.maximum(acc[:, 3:4], 1.0)
            sc = np.zeros(n)
            np.maximum.at(sc, of, np.abs(coord - cen[of]).max(1))
            sc_t = torch.as_tensor(sc, device=dev)
            be._allreduce(sc_t, dist.ReduceOp.MAX)
            return cen, sc_t.cpu().numpy()

        def orthonormal(G):
            """G [n][6][6] Gram matrices -> T [n][6][6] (columns: kept combinations scaled to unit norm, largest first; dropped ones
            zero) and the number kept: combinations that move nothing (a single line / node, clamped nodes) go, as in
            mg_build_level2 (rank_tol 1e-6 on the singular values)"""
            lam, V = np.linalg.eigh(G)
            lam, V = lam[:, ::-1], V[:, :, ::-1]
            keep = lam > 1e-12 * np.maximum(lam[:, :1], 1e-300)
            return V * (keep / np.sqrt(np.where(keep, lam, 1.0)))[:, None, :], keep.sum(1).astype(np.int32)

        cen, sc = centroids(agg, nagg)
        R = modes(cen, sc, agg)
        G = np.zeros((nagg, 6, 6))
        np.add.at(G, agg[own], np.einsum("ncj,nck->njk", R[own], R[own]))
        G_t = torch.as_tensor(G, device=dev)
        be._allreduce(G_t)
        T, arank = orthonormal(G_t.cpu().numpy())
        Z = np.einsum("ncj,njk->nck", R, T[agg])
        n2 = 6 * nagg
        self.ops.mg_set_level2(nagg, np.ascontiguousarray(agg, np.int32), np.ascontiguousarray(Z.reshape(self.nc, 18)))
        t0 = self._lap("  level 2: aggregates, modes (4 small all-reduces)", t0)
        # this rank's blocks of Z^T A1 Z, the global block pattern, the sum over the ranks
        sptr_l, cols_l, B_l = self.ops.mg_galerkin_blocks()
        t0 = self._lap("  level 2: Z^T A1 Z blocks (device)", t0)
        rows_l = np.repeat(np.arange(nagg, dtype=np.int64), np.diff(sptr_l))
        keys_l = rows_l * nagg + cols_l.astype(np.int64)
        every = [None] * part.world
        dist.all_gather_object(every, keys_l, group=part.group)
        allb = np.unique(np.concatenate(every))
        nslots = len(allb)
        Bg = torch.zeros(nslots, 36, dtype=torch.float64, device=dev)
        if len(keys_l):
            Bg.index_copy_(0, torch.as_tensor(np.searchsorted(allb, keys_l), device=dev), torch.as_tensor(B_l.reshape(-1, 36), device=dev))
        be._allreduce(Bg)
        Bsum = Bg.cpu().numpy()
        del Bg
        sptr_g = np.zeros(nagg + 1, dtype=np.int64)
        np.add.at(sptr_g, allb // nagg + 1, 1)
        sptr_g = np.cumsum(sptr_g)
        slot_agg = (allb % nagg).astype(np.int32)
        t0 = self._lap("  level 2: block pattern, all-reduce of %.1f MB" % (288e-6 * nslots), t0)
        max_dense = self.max_dense if self.max_dense else int(os.environ.get("FCVM_MG_MAX_AGGREGATES", "3000"))
        sparse = nagg > max_dense and os.environ.get("FCVM_MG_SPARSE", "1") != "0"
        nsuper, sagg, soff, Cm = 0, None, None, None
        if sparse:
            # super-aggregates: bins of f x f x f aggregate bins, at most FCVM_L2S_MAX_SUPER of them (as mg_build_sparse_levels)
            max_super = int(os.environ.get("FCVM_L2S_MAX_SUPER", str(min(max_dense, 2000))))
            ab = np.stack([allk // (nb[1] * nb[2]), (allk // nb[2]) % nb[1], allk % nb[2]], 1)
            for f in range(2, 64):
                q = ab // f
                m1, m2 = q[:, 1].max() + 1, q[:, 2].max() + 1
                u, sagg = np.unique((q[:, 0] * m1 + q[:, 1]) * m2 + q[:, 2], return_inverse=True)
                nsuper = len(u)
                if nsuper <= max_super:
                    break
            sagg = sagg.astype(np.int32)
            # C_g = Z_g^T R_S: the super-aggregate's rigid-body modes (about ITS centroid, orthonormalised over its level-1 nodes --
            # Gram matrices and the products with Z summed over the ranks) in the basis of its aggregates
            sof = sagg[agg].astype(np.int64)
            cen3, sc3 = centroids(sof, nsuper)
            R3 = modes(cen3, sc3, sof)
            GM = np.zeros((nsuper + nagg, 6, 6))
            np.add.at(GM, sof[own], np.einsum("ncj,nck->njk", R3[own], R3[own]))
            np.add.at(GM, nsuper + agg[own], np.einsum("nck,ncl->nkl", Z[own], R3[own]))
            GM_t = torch.as_tensor(GM, device=dev)
            be._allreduce(GM_t)
            GM = GM_t.cpu().numpy()
            T3, r3 = orthonormal(GM[:nsuper])
            Cm = np.ascontiguousarray(np.einsum("gkl,glm->gkm", GM[nsuper:], T3[sagg]))
            soff = np.concatenate([[0], np.cumsum(r3)]).astype(np.int64)
            t0 = self._lap("  level 2: super-aggregates (3 all-reduces)", t0)
        self.ops.mg_l2_finish(nagg, arank, sptr_g, slot_agg, Bsum, nsuper, sagg, soff, Cm)
        t0 = self._lap("  level 2: finish (%s)" % ("sparse level, A3 inverse, bound" if sparse else "dense inverse"), t0)
        self.nagg, self.n2 = nagg, n2
        self.r2 = TVec(torch.zeros(n2, dtype=torch.float64, device=dev))
        self.e2 = TVec(torch.zeros(n2, dtype=torch.float64, device=dev))
        self.l2_info = dict(aggregates=nagg, unknowns=n2, target=target, blocks=nslots, allreduce_bytes=288 * nslots, sparse=bool(sparse),
                            super_aggregates=int(nsuper), super_unknowns=int(soff[-1]) if sparse else 0)
        if os.environ.get("FCVM_VERBOSE") and part.rank == 0:
            print("fcvm dist: aggregate level %s" % self.l2_info, flush=True)

    # -- V-cycle ------------------------------------------------------------------------------
    def _spmv(self, level, x, y):
        self.ops.mg_spmv(level, x, y)
        (self.be.ex if level == 0 else self.ex1).interface_sum(y.t, 3)

    def _from_zero(self, level, ch, k, b, x, r, d, t):
        self.ops.mg_cheb_init(level, b, x, r, d, ch.c0)
        for j in range(1, k + 1):
            self._spmv(level, d, t)
            self.ops.mg_cheb_step(level, t, x, r, d, ch.cdd[j], ch.cdr[j], j < k)

    def _post(self, level, ch, k, x, r, d, t):
        self._spmv(level, d, t)
        self.ops.mg_cheb_step(level, t, x, r, d, 0.0, ch.c0, True)
        for j in range(1, k + 1):
            if j < k:
                self._spmv(level, d, t)
                self.ops.mg_cheb_step(level, t, x, r, d, ch.cdd[j], ch.cdr[j], True)
            else:
                self.ops.mg_cheb_step(level, None, x, r, d, 0.0, 0.0, False)

    def _coarse(self, r, partial=False, restricted=False, summed=False):
        """x1 = M1^-1 P^T (w0 r): weighted restriction of a consistent fine vector, level-1 / aggregate cycle.
        partial: r is already a per-rank PART of the fine vector (its sum over the ranks is the vector): restricted as it stands,
        the level-1 exchange then completes fine and coarse sums at once.  restricted: self.r1 already holds the rank's part."""
        be = self.be
        if restricted:
            pass
        elif partial:
            self.ops.mg_restrict(r, self.r1)
        else:
            self.ops.mg_restrict_weighted(r, TVec(self.w0), self.r1)
        if not summed:       # (summed: the caller restricted the shared corner nodes first and ran the exchange beside the rest)
            self.ex1.interface_sum(self.r1.t, 3)
        merge = self.merge_l1
        if self.fused_l1:
            self._coarse_fused_l1()
            return
        if merge:
            # k1s Chebyshev steps from zero, the last one's product A1 d left UNSUMMED in tk1: x is complete, rr1 still lacks it
            ch, k = self.ch1, self.k1s
            self.ops.mg_cheb_init(1, self.r1, self.x1, self.rr1, self.d1, ch.c0)
            for j in range(1, k):
                self._spmv(1, self.d1, self.t1)
                self.ops.mg_cheb_step(1, self.t1, self.x1, self.rr1, self.d1, ch.cdd[j], ch.cdr[j], True)
            self.ops.mg_spmv(1, self.d1, self.tk1)                                        # this rank's part of the last product
            self.ops.mg_cheb_step(1, None, self.x1, self.rr1, self.d1, 0.0, 0.0, False)   # x += d
            if hasattr(self.ops, "mg_l2_restrict_diff"):
                self.ops.mg_l2_restrict_diff(self.rr1, self.own1, self.tk1, self.r2)      # Z^T (own1 * rr1 - tk1): summed by the all-reduce
            else:
                torch.addcmul(-self.tk1.t, self.rr1.t, self.own1, out=self.t1.t)
                self.ops.mg_l2_restrict(self.t1, self.r2)
        else:
            self._from_zero(1, self.ch1, self.k1s, self.r1, self.x1, self.rr1, self.d1, self.t1)
            if hasattr(self.ops, "mg_l2_restrict_weighted"):
                self.ops.mg_l2_restrict_weighted(self.rr1, self.own1, self.r2)   # r2 = Z^T (own1 * r1): shared nodes count once (local part)
            else:
                torch.mul(self.rr1.t, self.own1, out=self.t1.t)       # owner-weighted: shared nodes count once
                self.ops.mg_l2_restrict(self.t1, self.r2)             # r2 = Z^T (.)  (local part)
        be._allreduce(self.r2.t)                              # replicated right-hand side of the aggregate level
        self.ops.mg_l2_solve(self.r2, self.e2)                # the same correction on every rank (replicated level: no gather)
        self.ops.mg_l2_prolong(self.e2, self.d1)
        if merge:
            # first post-smoothing product + the pending one, ONE exchange for both: rr1 -= sum over the ranks of (A1 d + tk1)
            ch, k = self.ch1, self.k1s
            self.ops.mg_spmv(1, self.d1, self.t1)
            self.t1.t.add_(self.tk1.t)
            self.ex1.interface_sum(self.t1.t, 3)
            self.ops.mg_cheb_step(1, self.t1, self.x1, self.rr1, self.d1, 0.0, ch.c0, True)
            for j in range(1, k + 1):
                if j < k:
                    self._spmv(1, self.d1, self.t1)
                    self.ops.mg_cheb_step(1, self.t1, self.x1, self.rr1, self.d1, ch.cdd[j], ch.cdr[j], True)
                else:
                    self.ops.mg_cheb_step(1, None, self.x1, self.rr1, self.d1, 0.0, 0.0, False)
        else:
            self._post(1, self.ch1, self.k1s, self.x1, self.rr1, self.d1, self.t1)

    def _l1_pass(self, din, dout, cdd, cdr, update_d, add_dout=False, t2=None):
        """one level-1 smoother pass on the library's 16-bit kernel with the step as epilogue; the rows of shared corner nodes are finished
        after the exchange of their raw products by one small kernel that sums and steps (no full-vector step kernel, no separate unpack)"""
        ex = self.ex1
        if getattr(self, "l1_split", False):
            self.ops.mg_l1_pass(din, dout, self.x1, self.rr1, t2, self.t1, self.defer1, cdd, cdr, update_d, add_dout, part=1)
            h = ex.begin(self.t1.t, 3)          # the shared rows' products are complete: pack and post while the interior slices run
            self.ops.mg_l1_pass(din, dout, self.x1, self.rr1, t2, self.t1, self.defer1, cdd, cdr, update_d, add_dout, part=2)
        else:
            self.ops.mg_l1_pass(din, dout, self.x1, self.rr1, t2, self.t1, self.defer1, cdd, cdr, update_d, add_dout)
            h = ex.begin(self.t1.t, 3)
        if ex.ranks:
            ex.end(h, finish=lambda recv: self.ops.unpack_sum_cheb(ex.inodes, ex.eptr, ex.src, self.t1, recv, din, dout, self.x1, self.rr1,
                                                                    cdd, cdr, update_d, add_dout))
        else:
            ex.end(h)

    def _coarse_fused_l1(self):
        """level 1 and below from the restricted, exchanged r1 (round 5: on k_spmv_l1_i16 with the Chebyshev step as epilogue -- what the
        one-GPU cycle runs -- instead of the FP32 sliced-ELL product plus a step kernel per pass).  Same algebra as the merge_l1 branch of
        _coarse: k1s steps from zero with the last product left unsummed (tk1), aggregate level, k1s steps after it with tk1 riding on the
        first exchange.  The direction ping-pongs between d1 and dn1 (a pass's rows still gather the old one)."""
        be, ch, k = self.be, self.ch1, self.k1s
        self.ops.mg_cheb_init(1, self.r1, self.x1, self.rr1, self.d1, ch.c0)
        din, dout = self.d1, self.dn1
        for j in range(1, k):
            self._l1_pass(din, dout, ch.cdd[j], ch.cdr[j], True)
            din, dout = dout, din
        self.ops.mg_l1_spmv(din, self.tk1)                                        # this rank's part of the last product
        self.ops.mg_cheb_step(1, None, self.x1, self.rr1, din, 0.0, 0.0, False)   # x += d
        self.ops.mg_l2_restrict_diff(self.rr1, self.own1, self.tk1, self.r2)      # Z^T (own1 * rr1 - tk1): summed by the all-reduce
        be._allreduce(self.r2.t)                              # replicated right-hand side of the aggregate level
        self.ops.mg_l2_solve(self.r2, self.e2)                # the same correction on every rank (replicated level: no gather)
        self.ops.mg_l2_prolong(self.e2, din)                  # the correction as the next direction
        # absorb it (x += d, r -= A1 d + the pending product), then k - 1 more steps of a fresh sequence and the closing x += d
        self._l1_pass(din, dout, 0.0, ch.c0, True, add_dout=(k == 1), t2=self.tk1)
        din, dout = dout, din
        for j in range(1, k):
            self._l1_pass(din, dout, ch.cdd[j], ch.cdr[j], True, add_dout=(j == k - 1))
            din, dout = dout, din

    def _cycle_lo(self, b, x):
        """form 2 from x1 = c0 Dinv b in self.d0:  x2 = x1 + P M1^-1 P^T (b - K~ x1);  x = x2 + c0 Dinv (b - K~ x2), K~ = the sum
        over the ranks of the 16-bit copies of their local matrices (one fine-level exchange: the second product's)"""
        be, c0 = self.be, self.weight * self.ch0.c0
        if self.fused_lo and self.lo_resid:
            # round 5: the residual after pre-smoothing as the epilogue of the smoother pass (one vector for the restriction to read instead of three)
            try:
                self.ops.spmv_lo_resid(self.d0, b, TVec(self.w0), self.r0)       # w0 b - K~_local x1
            except RuntimeError:       # no symmetric form of the 16-bit copy on this rank (small or graded mesh): the three-vector restriction
                self.lo_resid = False
        summed = False
        if self.fused_lo and self.lo_resid:
            if getattr(self, "l1_split", False):      # shared corner nodes first, their exchange beside the restriction of the others
                self.ops.mg_restrict(self.r0, self.r1, part=1)
                h = self.ex1.begin(self.r1.t, 3)
                self.ops.mg_restrict(self.r0, self.r1, part=2)
                self.ex1.end(h)
                summed = True
            else:
                self.ops.mg_restrict(self.r0, self.r1)
        else:
            self.ops.spmv_lo(self.d0, self.t0)
        if self.fused_lo:
            if not self.lo_resid:
                self.ops.mg_restrict_diff(b, TVec(self.w0), self.t0, self.r1)        # P^T (w0 b - K~_local x1), never written out
            self._coarse(None, restricted=True, summed=summed)
            self.ops.mg_prolong_add(self.x1, self.d0, self.x2)
            if be.split:       # boundary groups first, interior groups beside the messages (as DistBackend.spmv)
                self.ops.spmv_lo_part(self.x2, self.t0, 1)
                h = be.ex.begin(self.t0.t, 3)
                self.ops.spmv_lo_part(self.x2, self.t0, 2)
                be.ex.end(h)
            else:
                self.ops.spmv_lo(self.x2, self.t0)
                be.ex.interface_sum(self.t0.t, 3)
            if self.rz_rr is None:
                self._dots = torch.zeros(3, dtype=torch.float64, device=be.device)
                self.rz_rr = (self._dots[0:1], self._dots[2:3])
            self.ops.post_smooth(self.x2, b, self.t0, c0, be.w, x, self.rz_rr[0], self.rz_rr[1])
            return
        # round 4: the residual after pre-smoothing is only ever restricted, and the restriction is linear -- so the rank's
        # UNSUMMED product goes into it (b enters weighted by 1 / holders: a part of b) and the level-1 exchange that follows
        # the restriction anyway completes the sum over the ranks: one fine-level exchange less per cycle (mid-side nodes
        # interpolate from corners of their own element, so every part lands on coarse nodes of the rank that formed it)
        torch.addcmul(-self.t0.t, b.t, self.w0, out=self.r0.t)          # w0 b - K~_local x1
        self._coarse(self.r0, partial=True)
        self.ops.mg_prolong_add(self.x1, self.d0, self.x2)
        self.ops.spmv_lo(self.x2, self.t0)
        be.ex.interface_sum(self.t0.t, 3)
        torch.sub(b.t, self.t0.t, out=self.r0.t)
        self.ops.mg_cheb_init(0, self.r0, self.z0, self.r0, self.t0, c0)     # t0 = c0 Dinv (b - K~ x2)
        torch.add(self.x2.t, self.t0.t, out=x.t)

    def vcycle(self, b, x):
        """x = M^-1 b (consistent fine vectors)"""
        if self.form == 2:
            if self.fused_lo:
                self.ops.jacobi6f(b, self.weight * self.ch0.c0, self.d0)                  # x1 = c0 Dinv b (FP32 Dinv, as the other two uses)
            else:
                self.ops.mg_cheb_init(0, b, self.z0, b, self.d0, self.weight * self.ch0.c0)   # x1 = c0 Dinv b
            self._cycle_lo(b, x)
            return
        if self.additive:   # x = w c0 Dinv b + P M1^-1 P^T b : no fine-level SpMV / exchange in the preconditioner
            self.ops.mg_cheb_init(0, b, x, self.r0, self.d0, self.weight * self.ch0.c0)
            self._coarse(b)
            self.ops.mg_prolong_add(self.x1, self.d0, x)
            return
        self._from_zero(0, self.ch0, self.k0, b, x, self.r0, self.d0, self.t0)
        self._coarse(self.r0)
        self.ops.mg_prolong(self.x1, self.d0)
        self._post(0, self.ch0, self.k0, x, self.r0, self.d0, self.t0)

    def cg_update_and_vcycle(self, S, z, w, p, s, x, r):
        """the vector half of a Chronopoulos-Gear iteration (HipLocalOps.cg_update) and z = M^-1 r from its pre-smoothed start"""
        self.ops.cg_update(S, z, w, p, s, x, r, self.d0, self.weight * self.ch0.c0)
        self._cycle_lo(r, z)

    def update_and_vcycle(self, alpha, p, q, x, r, z):
        """additive cycle fused with the PCG update: x += alpha p, r -= alpha q, z = w c0 Dinv r + P M1^-1 P^T r
        (alpha: 1-element device tensor)"""
        if self.fused_lo:
            self.ops.pcg_update_xr_z0_f32(alpha, p, q, x, r, self.d0, self.weight * self.ch0.c0)
            self._cycle_lo(r, z)
            return
        self.ops.pcg_update_xr_z0(alpha, p, q, x, r, self.d0, self.weight * self.ch0.c0)
        if self.form == 2:      # d0 = x1 = c0 Dinv r is the pre-smoothing step
            self._cycle_lo(r, z)
            return
        self._coarse(r)
        self.ops.mg_prolong_add(self.x1, self.d0, z)


class DistModel:
    """Counterpart of solver.GpuModel for one rank of a partitioned mesh."""

    def __init__(self, part, backend, nocoord_local, material, fix_local, loads_local, external_loads_fn):
        self.ctx = backend
        self.part = part
        self.elNodes, self.nocoord = part.elNodes, np.ascontiguousarray(nocoord_local, dtype=np.float64)
        self.material = tuple(float(v) for v in material)
        self.fix = dict(fix_local)
        self.loads = loads_local
        self._ext_fn = external_loads_fn
        self._ext = None   # cached external load vector (device, exchanged), as solver.GpuModel; None: build it at the next calc_gsm
        self._follow = None   # follower pressure (GNLY): (FollowerPressure or None, its node numbers on the device, the static loads)
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
        # does ANY rank hold a pressure load?  Only pressure follows the deformed geometry (fcVM.py:858-883): without one the
        # external load vector is built once and kept on the device, as solver.GpuModel does -- no displacements back to the
        # host, no upload, no exchange per GNLY tangent.  The answer must be the same on every rank (the exchange is collective).
        hp = torch.tensor([1.0 if getattr(loads_local, "has_pressure", True) else 0.0], dtype=torch.float64, device=backend.device)
        backend._allreduce(hp, dist.ReduceOp.MAX)
        self.follower_loads = bool(hp.item() == 1.0)
        self.t_pattern = 0.0

    def calc_gsm(self, grav, disp=None, sig=None, pgp=None, Et_E=0.0, want_x=True):
        c = self.ctx
        xgp = c.empty(12 * self.ne) if want_x else None
        glv, modf, V = c.assemble(grav, disp=disp, sig=sig, pgp=pgp, Et_E=Et_E, xgp=xgp)
        follower = disp is not None and self.follower_loads
        if follower:
            # round 5: the static loads once (exchanged, kept on the device); the pressure from the displacements of this rank's loaded faces'
            # nodes alone -- gathered on the device, integrated on the host (O(loaded faces)), scattered into a zero vector and summed over the
            # shared nodes.  Rounds 1-4 took the whole displacement vector to the host at every tangent.  Every rank takes part in the
            # exchange, loaded or not.
            if self._follow is None:
                from .loads import FollowerPressure, split_follower
                static, _p = split_follower(self.loads)
                es = c.array(self._ext_fn(self.nocoord, static, None))
                c.ex.interface_sum(es.t, 3)
                fp = FollowerPressure(self.nocoord, self.loads) if getattr(self.loads, "has_pressure", False) else None
                idx = torch.as_tensor(fp.nodes.astype(np.int64), device=c.device) if fp is not None else None
                self._follow = (fp, idx, es)
            fp, idx, es = self._follow
            e = TVec(torch.zeros(3 * self.nn, dtype=torch.float64, device=c.device))
            if fp is not None and len(fp.nodes):
                d = disp.t.view(-1, 3).index_select(0, idx).cpu().numpy()
                f = torch.as_tensor(fp.loads(d), device=c.device)
                e.t.view(-1, 3).index_add_(0, idx, f)
            c.ex.interface_sum(e.t, 3)
            e.t.add_(es.t)
        elif self._ext is None:
            ext = self._ext_fn(self.nocoord, self.loads, None)
            # every rank takes part in the exchange, loaded or not (a rank-local "any load?" test would
            # leave the neighbours waiting)
            e = c.array(ext)
            c.ex.interface_sum(e.t, 3)
            self._ext = e          # (the driver resets it when the coordinates change: imperfection, set_coords)
        else:
            e = self._ext
        glv.t.add_(e.t)
        return glv, modf, V, xgp


def _local_loads(part, loads):
    """The rank's share of the load entities of the global model: every face / edge / vertex load goes to the LOWEST rank that
    holds all of its nodes (a face belongs to one element, so at least one rank does), renumbered to local nodes."""
    from .loads import Loads
    lut = np.zeros(int(part.gnodes.max()) + 2 if part.nn else 2, dtype=np.int64)
    lut[part.gnodes] = np.arange(1, part.nn + 1)

    def mine(nodes):
        if not len(nodes):
            held = np.zeros(0, dtype=bool)
        else:
            nodes = np.asarray(nodes, dtype=np.int64).reshape(len(nodes), -1)
            inr = nodes < len(lut)
            held = (inr & (lut[np.where(inr, nodes, 0)] > 0)).all(axis=1)
        every = [None] * part.world
        if part.world > 1:
            dist.all_gather_object(every, held, group=part.group)
        else:
            every = [held]
        first = np.full(len(held), -1)
        for r in range(part.world - 1, -1, -1):
            first[every[r]] = r
        if (first < 0).any():
            raise ValueError("a load entity whose nodes no rank holds together")
        return first == part.rank

    out = {}
    for nodes_name, vals_name in (("pfaces", "pvals"), ("vnodes", "vloads"), ("ufaces", "uloads"), ("edges", "eloads")):
        nodes, vals = getattr(loads, nodes_name), getattr(loads, vals_name)
        keep = mine(nodes)
        out[nodes_name], out[vals_name] = lut[nodes[keep]], vals[keep]
    return Loads(**out)


def spatial_element_order(elNodes, nocoord, bits=20):
    """Permutation of the elements along a Morton (Z-order) curve through their centroids: consecutive ranges of it are
    compact chunks of the mesh whatever the caller's element numbering (a mesh generator's locality is not guaranteed --
    a shuffled list cut into ranges would make every node an interface node).  Ties keep the caller's order (stable)."""
    el = np.asarray(elNodes, dtype=np.int64)
    xyz = np.asarray(nocoord, dtype=np.float64)
    cen = xyz[el[:, :4] - 1].mean(axis=1)
    lo, span = cen.min(axis=0), np.maximum(cen.max(axis=0) - cen.min(axis=0), 1e-300)
    q = np.minimum(((cen - lo) / span.max() * ((1 << bits) - 1)).astype(np.uint64), (1 << bits) - 1)

    def spread(v):      # bits of v to every third position (21-bit input)
        v = v & np.uint64(0x1FFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(key, kind="stable")


def interface_node_counts(elNodes, order, world):
    """Nodes of each rank's element range that another rank holds too (what every shared-node exchange moves), and the
    nodes each rank holds: the quality of an element-range partition, computed without any communication."""
    el = np.asarray(elNodes, dtype=np.int64)
    ne, nn = len(el), int(el.max())
    order = np.arange(ne) if order is None else np.asarray(order)
    holders = np.zeros(nn + 1, dtype=np.int32)
    sets = []
    for r in range(world):
        nodes = np.unique(el[order[r * ne // world:(r + 1) * ne // world]])
        holders[nodes] += 1
        sets.append(nodes)
    return [int((holders[nd] > 1).sum()) for nd in sets], [len(nd) for nd in sets]


def partition_model(elNodes, nocoord, material, fix, loads, device_index=0, device=None, group=None, precond="mg", order=None,
                    ops_factory=None, pcg_check_every=25, mg_options=None):
    """One rank's DistModel of the element-range partition of a global model (the 1 -> N split of north_star): elements
    [rank * ne / world, (rank + 1) * ne / world) of `order`, the nodes they touch, the supports and load entities that fall
    on them.  order: a permutation of the elements; None (default) = spatial_element_order (Morton curve through the element
    centroids: compact ranges for any numbering); "given" = the caller's element order as it stands.
    Collective: every rank of `group` calls it.
    ops_factory(part, nocoord_local, material, fix_local, element_offset) -> LocalOps: the rank's kernels (default: HipLocalOps on
    GPU `device_index`; the CPU tests pass their oracle-backed LocalOps and device = torch.device("cpu"))."""
    from .loads import Loads, external_loads
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    el = np.asarray(elNodes)
    ne = len(el)
    lo, hi = rank * ne // world, (rank + 1) * ne // world
    if order is None and world > 1:
        order = spatial_element_order(el, nocoord)
    sel = np.arange(lo, hi) if (order is None or isinstance(order, str)) else np.asarray(order)[lo:hi]
    part = Partition(el[sel], group=group)
    xyz_loc = np.ascontiguousarray(np.asarray(nocoord, dtype=np.float64)[part.gnodes - 1])
    gd = np.array(sorted(fix), dtype=np.int64)
    here, ld = part.local_dofs_of(gd)
    fix_loc = {int(d): float(fix[int(g)]) for d, g in zip(ld, gd[here])}
    if ops_factory is None:
        ops = HipLocalOps(part, xyz_loc, material, fix_loc, device_index, element_offset=lo)
        device = device if device is not None else torch.device("cuda", device_index)
    else:
        ops = ops_factory(part, xyz_loc, material, fix_loc, lo)
    be = DistBackend(part, ops, device, pcg_check_every=pcg_check_every)
    be.element_ids = np.asarray(sel, dtype=np.int64)    # crip (fcVM.py:2160 argmax over csr) in the caller's numbering
    if precond == "mg":
        be.enable_multigrid(xyz_loc, **(mg_options or {}))
    model = DistModel(part, be, xyz_loc, material, fix_loc, _local_loads(part, loads if loads is not None else Loads()),
                      external_loads)
    model.elements = sel
    return model


class HipLocalOps:
    """LocalOps on libfcvm_hip: the rank's kernels run on torch's current HIP stream, operating on
    the raw device pointers of the torch tensors.  No fallback: needs the library and a GPU."""

    def __init__(self, part, nocoord_local, material, fix_local, device_index, element_offset=0):
        from .device import Context
        self.c = Context(device_index)
        self.c.set_stream(torch.cuda.current_stream(device_index).cuda_stream)
        self.c.set_precond("bj")  # mg_enable() switches to "hierarchy only": the V-cycle is driven by DistMG
        self.c.set_mesh(part.elNodes, nocoord_local)
        self.c.set_material(*material)
        self.c.set_fix(fix_local)
        self.c.finish_setup()
        self.element_offset = element_offset
        self.local_mg = False

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
        if self.local_mg and self.c.precond_info()["l2_aggregates"] == 0:
            self.c.build_level2()   # single rank: the library's own three-level cycle (as device.Context.assemble)
        return vol.value

    def enable_local_multigrid(self):
        self.c.set_precond("mg")
        self.local_mg = True

    def solve_local(self, f, u, rtol, maxit):
        _, info = self.c.solve(self._p(f), self._p(u), rtol=rtol, maxit=maxit)
        return info

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

    def pack(self, idx, ncomp, t, out):
        from ._lib import call
        call("fcvm_pack_nodes", self.c.h, idx.numel(), idx.data_ptr(), int(ncomp), t.data_ptr(), out.data_ptr())

    def unpack_sum(self, inodes, eptr, src, ncomp, t, recv):
        from ._lib import call
        call("fcvm_unpack_sum", self.c.h, inodes.numel(), inodes.data_ptr(), eptr.data_ptr(), src.data_ptr(), int(ncomp),
             t.data_ptr(), recv.data_ptr())

    def assemble_buckling(self, sig, sigma):
        from ._lib import call
        call("fcvm_assemble_buckling", self.c.h, sig.ptr, float(sigma))

    def spmv_geometric(self, x, y):
        from ._lib import call
        call("fcvm_spmv_geometric", self.c.h, x.ptr, y.ptr)

    def set_coords(self, nocoord_local):
        self.c.set_coords(np.ascontiguousarray(nocoord_local, dtype=np.float64))

    def pcg_scalars(self, phase, S, t_in):
        from ._lib import call
        call("fcvm_dist_pcg_scalars", self.c.h, int(phase), S.data_ptr(), t_in.data_ptr())

    def wdot(self, w, x, y, out):
        """out[0] = sum w x y (torch tensors on this rank's device)"""
        from ._lib import call
        call("fcvm_vec_wdot_dev", self.c.h, x.numel(), w.data_ptr(), x.data_ptr(), y.data_ptr(), out.data_ptr())

    # -- multigrid building blocks (fcvm_set_precond mode 2 builds the hierarchy at assemble) --
    def mg_enable(self):
        from ._lib import call
        call("fcvm_set_precond", self.c.h, 2, 1, 1, 4.0, 8.0)

    def mg_coarse_map(self):
        from ._lib import call
        nc = self.c.precond_info()["coarse_nodes"]
        out = np.empty(nc, np.int32)
        call("fcvm_mg_coarse_map", self.c.h, out.ctypes.data)
        return out

    def mg_coarse_nodes(self):
        from ._lib import call
        nc = self.c.precond_info()["coarse_nodes"]
        coord, fm = np.empty((nc, 3)), np.empty(nc, np.uint8)
        call("fcvm_mg_coarse_nodes", self.c.h, coord.ctypes.data, fm.ctypes.data)
        return coord, fm

    def mg_spmv(self, level, x, y):
        from ._lib import call
        call("fcvm_spmv_level", self.c.h, level, x.ptr, y.ptr)

    def spmv_lo(self, x, y):
        """y = K~ x with the 16-bit copy of the rank's matrix (the smoother passes of the multiplicative cycle)"""
        from ._lib import call
        call("fcvm_spmv_lo", self.c.h, x.ptr, y.ptr)

    def mg_get_diag(self, level, dblk):
        from ._lib import call
        call("fcvm_mg_get_diag_blocks", self.c.h, level, dblk.ptr)

    def mg_set_diag(self, level, dblk):
        from ._lib import call
        call("fcvm_mg_set_diag_blocks", self.c.h, level, dblk.ptr)

    def mg_cheb_init(self, level, b, x, r, d, c0):
        from ._lib import call
        call("fcvm_mg_cheb_init", self.c.h, level, b.ptr, x.ptr, r.ptr, d.ptr, float(c0))

    def mg_cheb_step(self, level, t, x, r, d, cdd, cdr, update_d):
        from ._lib import call
        call("fcvm_mg_cheb_step", self.c.h, level, t.ptr if t is not None else None, x.ptr, r.ptr, d.ptr, float(cdd),
             float(cdr), 1 if update_d else 0)

    def mg_restrict(self, r0, r1, part=0):
        from ._lib import call
        call("fcvm_mg_restrict_part", self.c.h, r0.ptr, r1.ptr, int(part))

    def mg_restrict_weighted(self, r0, w, r1):
        from ._lib import call
        call("fcvm_mg_restrict_weighted", self.c.h, r0.ptr, w.ptr, r1.ptr)

    def mg_prolong_add(self, e1, z0, z):
        from ._lib import call
        call("fcvm_mg_prolong_add", self.c.h, e1.ptr, z0.ptr, z.ptr)

    def pcg_update_xr_z0(self, alpha, p, q, x, r, z0, weight):
        from ._lib import call
        call("fcvm_pcg_update_xr_z0", self.c.h, alpha.data_ptr(), p.ptr, q.ptr, x.ptr, r.ptr, z0.ptr, float(weight))

    # round 4: fused forms for the cycle of form 2, all with the six-entry FP32 inverse diagonal blocks (one smoother S before
    # and after the coarse correction); DistMG uses them together or not at all (`fused_lo`)
    def pcg_update_xr_z0_f32(self, alpha, p, q, x, r, z0, weight):
        from ._lib import call
        call("fcvm_pcg_update_xr_z0_f32", self.c.h, alpha.data_ptr(), p.ptr, q.ptr, x.ptr, r.ptr, z0.ptr, float(weight))

    def jacobi6f(self, r, weight, z):
        from ._lib import call
        call("fcvm_jacobi6f", self.c.h, r.ptr, float(weight), z.ptr)

    def mg_restrict_diff(self, r0, w, sub, r1):
        """r1 = P^T (w * r0 - sub)"""
        from ._lib import call
        call("fcvm_mg_restrict_diff", self.c.h, r0.ptr, w.ptr, sub.ptr, r1.ptr)

    def post_smooth(self, x2, b, t, c0, ow, z, bz, bb):
        """z = x2 + c0 Dinv (b - t);  bz[0] = sum ow b z,  bb[0] = sum ow b b  (ow, bz, bb: torch tensors)"""
        from ._lib import call
        call("fcvm_dist_post_smooth", self.c.h, x2.ptr, b.ptr, t.ptr, float(c0), ow.data_ptr(), z.ptr, bz.data_ptr(), bb.data_ptr())

    def cg_update(self, S, z, w, p, s, x, r, z0, weight):
        """p = z + beta p, s = w + beta s, x += alpha p, r -= alpha s, z0 = weight Dinv r (alpha = S[5], beta = S[6], torch tensor S)"""
        from ._lib import call
        call("fcvm_dist_cg_update", self.c.h, S.data_ptr(), z.ptr, w.ptr, p.ptr, s.ptr, x.ptr, r.ptr, z0.ptr, float(weight))

    def mg_l2_restrict_weighted(self, r1, wgt, r2):
        from ._lib import call
        call("fcvm_mg_l2_restrict_weighted", self.c.h, r1.ptr, wgt.data_ptr(), r2.ptr)

    def mg_l2_restrict_diff(self, r1, wgt, sub, r2):
        """r2 = Z^T (wgt * r1 - sub)"""
        from ._lib import call
        call("fcvm_mg_l2_restrict_diff", self.c.h, r1.ptr, wgt.data_ptr(), sub.ptr, r2.ptr)

    def spmv_dot(self, x, y, out):
        """y = K_local x and out[0] = x . y over the local rows (the rank's share of x^T K x for a consistent x)"""
        from ._lib import call
        call("fcvm_spmv_dot", self.c.h, x.ptr, y.ptr, out.data_ptr())

    def mg_l1_16(self):
        """does the library keep the 16-bit copy of the level-1 matrix (FCVM_MG_L1_16, default on)?"""
        return os.environ.get("FCVM_MG_L1_16", "1") != "0"

    def mg_l1_pass(self, din, dout, x, r, t2, t_out, defer, cdd, cdr, update_d, add_dout, part=0):
        from ._lib import call
        call("fcvm_mg_l1_pass_part", self.c.h, din.ptr, dout.ptr, x.ptr, r.ptr, t2.ptr if t2 is not None else None, t_out.ptr, defer.data_ptr(),
             float(cdd), float(cdr), 1 if update_d else 0, 1 if add_dout else 0, int(part))

    def mg_l1_set_interface(self, nodes):
        """the level-1 nodes other ranks hold too (level-1 numbering): mg_l1_pass(part = 1) launches the slices with a row of one of them, part = 2
        the others.  Returns (boundary slices, interior slices)"""
        import ctypes as C
        from ._lib import call
        nodes = np.ascontiguousarray(nodes, np.int64)
        out = (C.c_int64 * 2)()
        call("fcvm_mg_l1_set_interface", self.c.h, len(nodes), nodes.ctypes.data if len(nodes) else None, out)
        return int(out[0]), int(out[1])

    def mg_l1_spmv(self, x, y):
        from ._lib import call
        call("fcvm_mg_l1_spmv", self.c.h, x.ptr, y.ptr)

    def unpack_sum_cheb(self, inodes, eptr, src, traw, recv, din, dout, x, r, cdd, cdr, update_d, add_dout):
        from ._lib import call
        call("fcvm_unpack_sum_cheb", self.c.h, inodes.numel(), inodes.data_ptr(), eptr.data_ptr(), src.data_ptr(), traw.ptr, recv.data_ptr(),
             din.ptr, dout.ptr, x.ptr, r.ptr, float(cdd), float(cdr), 1 if update_d else 0, 1 if add_dout else 0)

    def set_interface_nodes(self, nodes):
        """the local nodes other ranks hold too: the groups of the window format with a row of one of them are launched first by the *_part calls.
        Returns (boundary groups, interior groups); (0, 0) where the storage form has no lists"""
        import ctypes as C
        from ._lib import call
        nodes = np.ascontiguousarray(nodes, np.int64)
        out = (C.c_int64 * 2)()
        call("fcvm_set_interface_nodes", self.c.h, len(nodes), nodes.ctypes.data if len(nodes) else None, out)
        return int(out[0]), int(out[1])

    def spmv_dot_part(self, x, y, out, part):
        from ._lib import call
        call("fcvm_spmv_dot_part", self.c.h, x.ptr, y.ptr, out.data_ptr(), int(part))

    def spmv_lo_resid(self, x, r, w, out):
        """out = w r - K~_local x (16-bit copy): the smoother pass with the weighted residual as epilogue"""
        from ._lib import call
        call("fcvm_spmv_lo_resid", self.c.h, x.ptr, r.ptr, w.ptr, out.ptr)

    def spmv_lo_part(self, x, y, part):
        from ._lib import call
        call("fcvm_spmv_lo_part", self.c.h, x.ptr, y.ptr, int(part))

    def mg_set_level2(self, nagg, agg, Z):
        """transfer operators of the aggregate level: global aggregate of every level-1 node, Z [nc][3][6] (six slots per aggregate)"""
        from ._lib import call
        aoff = 6 * np.arange(nagg + 1, dtype=np.int64)
        call("fcvm_mg_set_level2", self.c.h, nagg, 6 * nagg, agg.ctypes.data, aoff.ctypes.data, Z.ctypes.data, None, 3, 8.0)

    def mg_galerkin_blocks(self):
        """this rank's 6 x 6 blocks of Z^T A1 Z (block-CSR over the global aggregates): (row pointer [nagg + 1], column aggregate, values)"""
        import ctypes as C
        from ._lib import call
        n = C.c_int64(0)
        call("fcvm_mg_l2_galerkin", self.c.h, C.byref(n))
        nagg = self.c.precond_info()["l2_aggregates"]
        sptr, cols, B = np.empty(nagg + 1, np.int64), np.empty(max(n.value, 1), np.int32), np.empty(max(n.value, 1) * 36)
        call("fcvm_mg_l2_galerkin_get", self.c.h, sptr.ctypes.data, cols.ctypes.data, B.ctypes.data)
        return sptr, cols[:n.value], B[:n.value * 36]

    def mg_l2_finish(self, nagg, arank, sptr_g, slot_agg, B, nsuper, sagg, soff, Cm):
        """the replicated level from the summed blocks (fcvm_mg_l2_finish): dense inverse (nsuper = 0) or sparse level + super-aggregates"""
        from ._lib import call
        arank, sptr_g, slot_agg = (np.ascontiguousarray(arank, np.int32), np.ascontiguousarray(sptr_g, np.int64),
                                   np.ascontiguousarray(slot_agg, np.int32))
        B = np.ascontiguousarray(B, np.float64)
        if nsuper:
            sagg, soff, Cm = np.ascontiguousarray(sagg, np.int32), np.ascontiguousarray(soff, np.int64), np.ascontiguousarray(Cm, np.float64)
        # sparse form: 2 smoothing steps per sweep, 2 super-aggregate corrections (the one-GPU library's 8 / 2 cost 25 launches per cycle on EVERY
        # rank; 13 k aggregates of the 6.0 M plate, gpurun_out/r05r: 8 / 2 -> 49 PCG iterations, 4 / 2 -> 54, 2 / 2 -> 49, 2 / 3 -> 48, 1 / 2 -> 55)
        call("fcvm_mg_l2_finish", self.c.h, int(nagg), arank.ctypes.data, len(slot_agg), sptr_g.ctypes.data, slot_agg.ctypes.data, B.ctypes.data,
             int(nsuper), sagg.ctypes.data if nsuper else None, soff.ctypes.data if nsuper else None, Cm.ctypes.data if nsuper else None, 2, 2)

    def mg_l2_restrict(self, r1, r2):
        from ._lib import call
        call("fcvm_mg_l2_restrict", self.c.h, r1.ptr, r2.ptr)

    def mg_l2_solve(self, r2, e2):
        """e2 = M2^-1 r2 on the replicated aggregate level (dense inverse from its lower triangle, or one cycle of the sparse level)"""
        from ._lib import call
        call("fcvm_mg_l2_solve", self.c.h, r2.ptr, e2.ptr)

    def mg_l2_prolong(self, e2, d1):
        from ._lib import call
        call("fcvm_mg_l2_prolong", self.c.h, e2.ptr, d1.ptr)

    def mg_prolong(self, e1, d0):
        from ._lib import call
        call("fcvm_mg_prolong", self.c.h, e1.ptr, d0.ptr)
