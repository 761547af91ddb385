"""BASELINE.json configs[3] and configs[4] on one MI355X (and, for [3], on ranks sharing GPU 0).

  configs[3]  Embankment_with_Ditch_Example.inp, mesh refined to the ~500k class (3 uniform refinements of the
              reference's 659-element mesh = 337,408 tet10 / 1.45 M dof), full elastoplastic 30-step collapse
              analysis; the 1 -> 8 GPU element partition is rehearsed with 2 and 3 ranks on the one GPU.
  configs[4]  synthetic cube, 8 M tet10 (110^3 cells = 7,986,000), von Mises plasticity, displacement control
              (prescribed top displacement -> reaction-force output, fcVM.py:1169-1177, 1522-1523).

No oracle at these sizes: the checks are the reference's own result for the same model (collapse plateau of the
shipped .out), mesh convergence, and size-independent properties (equilibrium of the converged steps, self-equilibrated
internal forces, reaction = sum of internal forces at the moved dofs, symmetry of K)."""
import os

import numpy as np
import pytest

from conftest import spawn_sharing_gpu  # noqa: E402

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _embankment_run(refine, nstep=None, rtol=1e-10):
    from fcvm_hip import examples
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel, RiksDriver, calc_disp
    case = examples.embankment(refine=refine)
    ctl = dict(case["ctl"]) if nstep is None else dict(case["ctl"], nstep=nstep)
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads())
    st = {}
    res = calc_disp(m, ctl, rtol=rtol, stats=st)
    return case, m, res, st, RiksDriver


def test_config3_embankment_337k_full_collapse_analysis(embankment):
    """configs[3] on one GPU: all 30 arc-length steps of the reference's control file on the 337k-element refinement."""
    g, _ = embankment
    ref_plateau = float(g["out_table"][-1, 4])            # 0.503: collapse load factor of the reference's 659-element run
    case, m, res, st, _ = _embankment_run(3)
    c = m.ctx
    assert m.ne == 337408 and m.ndof == 3 * 482409
    lout, un = np.asarray(res["lout"]), np.asarray(res["un"])
    assert len(lout) == 31 and not res["fail"]
    # gravity-driven collapse: the load factor saturates while the displacement keeps growing
    assert np.all(np.diff(un) > 0.0)
    assert lout[-1] - lout[-6] < 0.01 * lout[-1] and un[-1] > 1.3 * un[-6]
    # upper-bound character of the displacement FE solution: the finer mesh collapses at or below the coarse mesh's load
    assert 0.80 * ref_plateau < lout[-1] <= ref_plateau * 1.001, (lout[-1], ref_plateau)
    assert res["peeq"].get().max() > 0.05
    # total load = weight of the soil body (the shipped .out prints -1.06e+06 for the same geometry)
    assert res["loadsum"][2] == pytest.approx(-1.06e6, rel=5e-3)
    assert abs(res["loadsum"][0]) < 1e-6 * abs(res["loadsum"][2]) and abs(res["loadsum"][1]) < 1e-6 * abs(res["loadsum"][2])
    # K symmetric on the unstructured (Morton row order) mesh
    rng = np.random.default_rng(5)
    dx, dy = c.array(rng.standard_normal(m.ndof)), c.array(rng.standard_normal(m.ndof))
    kx, ky = c.spmv(dx, c.empty(m.ndof)), c.spmv(dy, c.empty(m.ndof))
    assert c.dot(dy, kx) == pytest.approx(c.dot(dx, ky), rel=1e-10)
    # the multigrid solver did the work it is there for
    assert st["solves"] > 100 and st["pcg_iterations"] / st["solves"] < 250
    print("config3: 337k collapse load factor %.5f (reference 659 elements %.5f), %d Newton iterations, %.1f PCG its/solve"
          % (lout[-1], ref_plateau, st["newton_iterations"], st["pcg_iterations"] / st["solves"]))
    m.ctx.close()
    # mesh convergence 659 (the reference's own .out) -> 5,272 -> 42,176 -> 337,408 elements: the collapse load factor
    # decreases monotonically and the steps shrink
    lf = [ref_plateau]
    for refine in (1, 2):
        _, m2, r2, _, _ = _embankment_run(refine)
        assert len(r2["lout"]) == 31
        lf.append(float(r2["lout"][-1]))
        m2.ctx.close()
    lf.append(float(lout[-1]))
    d = -np.diff(lf)
    print("config3 mesh convergence: load factors", lf)
    assert np.all(d > -1e-4) and d[2] < d[0] + 1e-4, lf


def test_config3_equilibrium_of_a_converged_step():
    """The out-of-balance force the driver reports is the real one: recomputed from the final stresses with an
    independent internal-force evaluation (elastic trial from the converged state with du = 0)."""
    from fcvm_hip import examples
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel, RiksDriver
    case = examples.embankment(refine=3)
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads())
    c = m.ctx
    d = RiksDriver(m, dict(case["ctl"], nstep=30))
    for _ in range(4):
        err = d.start_step()
        while err > d.error_max:
            err = d.iteration()
            assert d.iterat <= d.iterat_max
        lam = d.lbd[d.step + 1]
        # independent evaluation: stresses sig_new are in equilibrium with lam * glv on the free dofs
        r = c.axpby(lam, d.glv, -1.0, d.qin)
        r = c.mul(m.fixdof, r)
        assert c.norm(r) / d.qnorm <= d.error_max
        # internal forces are self-equilibrated (sum over all nodes of B^T sigma = 0 in every direction)
        q = d.qin.get().reshape(-1, 3)
        assert np.abs(q.sum(0)).max() < 1e-9 * np.abs(q).sum()
        d.end_step()
    assert int(d.pgp.get().sum()) > 0
    m.ctx.close()


def _worker_emb337(rank, world, port, out, nstep):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fcvm_hip import examples
        from fcvm_hip.dist import DistBackend, DistModel, HipLocalOps, Partition
        from fcvm_hip.loads import Loads, external_loads
        from fcvm_hip.solver import calc_disp
        torch.cuda.set_device(0)
        case = examples.embankment(refine=3)
        el, xyz = case["elNodes"], case["nocoord"]
        ne = len(el)
        lo, hi = rank * ne // world, (rank + 1) * ne // world
        part = Partition(el[lo:hi])
        xyz_loc = xyz[part.gnodes - 1]
        gd = np.array(sorted(case["fix"]), dtype=np.int64)
        here, ld = part.local_dofs_of(gd)
        fix_loc = {int(d): case["fix"][int(g)] for d, g in zip(ld, gd[here])}
        ops = HipLocalOps(part, xyz_loc, case["material"], fix_loc, 0, element_offset=lo)
        be = DistBackend(part, ops, torch.device("cuda", 0))
        be.enable_multigrid(xyz_loc)
        model = DistModel(part, be, xyz_loc, case["material"], fix_loc, Loads(), external_loads)
        res = calc_disp(model, dict(case["ctl"], nstep=nstep), rtol=1e-10, maxit=4000)
        payload = dict(gnodes=part.gnodes, disp=res["displacements"], lout=res["lout"], un=res["un"], crip=res["crip"],
                       pcg=res["stats"]["pcg_iterations"], solves=res["stats"]["solves"], ne=hi - lo)
        got = [None] * world
        dist.all_gather_object(got, payload)
        if rank == 0:
            torch.save(got, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_config3_embankment_337k_element_partition_matches_one_rank(tmp_path, world):
    """configs[3]'s 1 -> N element partition, rehearsed with N ranks sharing GPU 0 (gloo control plane, host-staged
    shared-node exchange; on the 8-GPU node the same code runs over RCCL): first load steps of the 337k-element model
    against the one-rank run, north_star tolerance 1e-6."""
    import socket
    import torch
    import torch.multiprocessing as mp
    nstep = 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r.pt")
    spawn_sharing_gpu(_worker_emb337, (world, port, out, nstep), world)
    got = torch.load(out, weights_only=False)
    _, m, ref, _, _ = _embankment_run(3, nstep=nstep)
    m.ctx.close()
    assert sum(g["ne"] for g in got) == 337408
    for g in got:
        assert len(g["lout"]) == len(ref["lout"]) == nstep + 1
        assert rel(g["lout"], ref["lout"]) < 1e-6 and rel(g["un"], ref["un"]) < 1e-6
        assert np.array_equal(g["crip"], ref["crip"])
        d = ref["displacements"].reshape(-1, 3)[g["gnodes"] - 1].reshape(-1)
        assert rel(g["disp"], d) < 1e-6
    print("config3 x%d ranks: %.1f PCG its/solve" % (world, got[0]["pcg"] / got[0]["solves"]))


def test_config4_cube_8m_von_mises_displacement_control():
    """configs[4] on one GPU: 110^3 cells = 7,986,000 tet10 / 32.4 M dof, base clamped, top face moved down by a
    prescribed displacement (movdof), von Mises plasticity; three load steps into the plastic range, driven through
    the same start_step / iteration / restart / end_step moves as calcDisp."""
    from fcvm_hip import meshgen
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel, RiksDriver
    n, L = 110, 100.0
    el, xyz = meshgen.cube_tet10(n, n, n, L, L, L)
    assert len(el) == 7986000
    base = meshgen.nodes_on_plane(xyz, 2, 0.0)
    top = meshgen.nodes_on_plane(xyz, 2, L)
    fix = {3 * int(nd) + k: 0.0 for nd in base for k in range(3)}
    delta = -0.08                      # mm at load factor 1: nominal strain 8e-4 -> 168 N/mm2 elastic, yield at 120
    for nd in top:
        fix[3 * int(nd) + 2] = delta
    ctl = dict(sig_yield=120.0, grav_x=0.0, grav_y=0.0, grav_z=0.0, nstep=3, iterat_max=25, error_max=5.0e-3, relax=1.2,
               scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25, Et_E=0.02, target_LF=10.0,
               csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLN", maxImp="0.0", ev1="1.0", ev2="0.0")
    m = GpuModel(el, xyz, (210000.0, 0.3, 7.8e-6), fix, Loads())
    del el
    c = m.ctx
    assert m.moving and m.ne == 7986000
    st = {}
    d = RiksDriver(m, ctl, stats=st)
    # elastic reaction on the moved boundary: own evaluation of sum(movdof * q(ue)) with the yield stress out of reach
    ne = m.ne
    q = c.empty(m.ndof)
    c.update_stress_load(c.array(np.full(4 * ne, 1e30)), None, d.ue, c.zeros(24 * ne), c.empty(24 * ne), c.empty(24 * ne), q, 0.0,
                         False, c.empty(4 * ne, np.uint8))
    r_el = c.dot(m.movdof, q)
    del q
    c.trim_pool()
    # uniaxial estimate E A delta / L (compression), stiffened by the clamped base: within 15 %
    assert r_el < 0.0 and abs(r_el) == pytest.approx(210000.0 * L * L * abs(delta) / L, rel=0.15)
    done = 0
    while done < 3 and not d.mrr:
        err = d.start_step()
        while err > d.error_max and not d.mrr:
            err = d.iteration()
            if d.iterat > d.iterat_max:
                d.restart()
                err = d.error
        assert not d.mrr, "maximum restarts reached"
        # equilibrium on the free dofs (no external load: the residual is the internal force itself)
        assert c.norm(c.mul(m.fixdof, d.qin)) / d.qnorm <= d.error_max
        # internal forces self-equilibrated over the whole mesh; reaction at the top = - reaction at the base
        qh = d.qin.get().reshape(-1, 3)
        assert np.abs(qh.sum(0)).max() < 1e-9 * np.abs(qh).sum()
        rtop = qh[top, 2].sum()
        assert rtop == pytest.approx(-qh[base, 2].sum(), rel=4 * d.error_max)
        assert not d.end_step()
        # reaction-force output of displacement control = sum of the internal forces at the moved dofs (fcVM.py:1522-1523)
        assert d.rfl[-1] == pytest.approx(rtop, rel=1e-11)
        done += 1
    res = d.result()
    lbd, rfl = np.asarray(d.lbd), np.asarray(res["lout"])
    assert len(rfl) == 4 and np.array_equal(rfl, np.asarray(d.rfl))
    # reactions: on (or, with the first local yielding at the clamped edges, just below) the elastic line at first,
    # clearly below it once the block yields, still growing in magnitude (hardening)
    assert abs(lbd[1] * r_el) * 0.97 < abs(rfl[1]) <= abs(lbd[1] * r_el) * (1 + 1e-9)
    nplast = int(d.pgp.get().sum())
    assert nplast > 0.25 * 4 * ne
    assert abs(rfl[3]) < 0.95 * abs(lbd[3] * r_el) and abs(rfl[3]) > abs(rfl[2]) > abs(rfl[1])
    assert abs(rfl[3]) > 0.9 * 120.0 * L * L        # the block carries at least its uniaxial yield load
    print("config4: 8M tet10, reactions", rfl, "load factors", lbd, "plastic GPs %d" % nplast,
          "Newton iterations %d, PCG its/solve %.1f" % (st["newton_iterations"], st["pcg_iterations"] / st["solves"]))
    m.ctx.close()
