"""Two ranks sharing GPU 0 (gloo for the control plane, host-staged exchange) run the distributed
driver with the real HIP kernels (HipLocalOps) and must reproduce the single-GPU analysis of the
same global mesh.  On the 8-GPU node the same code runs with backend nccl (RCCL), one GPU each."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import spawn_sharing_gpu  # noqa: E402

pytestmark = pytest.mark.gpu

CELLS = 4
CTL = dict(sig_yield=120.0, grav_x=0.0, grav_y=0.0, grav_z=-9.81, nstep=3, iterat_max=20, error_max=1e-5, relax=1.2,
           scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25, Et_E=0.02, target_LF=10.0,
           csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLN", maxImp="0.0", ev1="1.0", ev2="0.0")
MAT = (210000.0, 0.3, 7.8e-6)
TRACTION = (20.0, 0.0, 200.0)


def _worker(rank, world, port, out, precond):
    if precond == "mg-unfused":     # the cycle's end and the inner products as separate kernels (the rounds-1-3 sequence)
        os.environ["FCVM_DIST_FUSED_LO"] = "0"
        os.environ["FCVM_DIST_MERGE_L1"] = "0"      # ... and every level-1 product exchanged on its own
        precond = "mg"
    os.environ["FCVM_PRECOND"] = precond
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fcvm_hip.dist_bench import build_rank_model
        from fcvm_hip.solver import calc_disp
        torch.cuda.set_device(0)
        model, be, ops = build_rank_model(rank, world, CELLS, MAT, TRACTION, 0, torch.device("cuda", 0), length=10.0)
        res = calc_disp(model, CTL, rtol=1e-12, maxit=4000)
        payload = dict(gnodes=model.part.gnodes, disp=res["displacements"], lout=res["lout"], un=res["un"],
                       crip=res["crip"], peeq=res["peeq"].get(), pcg=res["stats"]["pcg_iterations"],
                       solves=res["stats"]["solves"])
        got = [None] * world
        dist.all_gather_object(got, payload)
        if rank == 0:
            torch.save(got, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precond", ["mg", "mg-unfused", "bj"])
def test_two_ranks_on_one_gpu_match_single_gpu(tmp_path, precond):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r.pt")
    spawn_sharing_gpu(_worker, (2, port, out, precond), 2)
    got = torch.load(out, weights_only=False)
    from fcvm_hip import meshgen
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    c = CELLS
    el, xyz = meshgen.cube_tet10(c, c, 2 * c, 10.0, 10.0, 20.0)
    fix = {3 * int(n) + k: 0.0 for n in meshgen.nodes_on_plane(xyz, 2, 0.0) for k in range(3)}
    faces = meshgen.boundary_faces_on_plane(el, xyz, 2, 20.0)
    tr = np.array(TRACTION) * [0.5, 0.5, 1.0]   # build_rank_model scales the shear part by 1 / world
    ref = run_analysis(el, xyz, MAT, fix, Loads(ufaces=faces, uloads=np.tile(tr, (len(faces), 1))), CTL, rtol=1e-12)
    assert ref["peeq"].max() > 0
    rel = lambda a, b: np.abs(np.asarray(a) - b).max() / np.abs(b).max()  # noqa: E731
    ne_half = len(el) // 2
    for r, g in enumerate(got):
        assert rel(g["lout"], ref["lout"]) < 1e-8 and rel(g["un"], ref["un"]) < 1e-8
        d = ref["displacements"].reshape(-1, 3)[g["gnodes"] - 1].reshape(-1)
        assert rel(g["disp"], d) < 1e-8
    # the slab partition of the column == the first / second half of the global element list up to
    # a permutation of elements inside the slabs: compare sorted plastic strains
    per_solve = got[0]["pcg"] / got[0]["solves"]
    assert per_solve < 90 if precond.startswith("mg") else per_solve > 100, per_solve
    allp = np.sort(np.concatenate([g["peeq"] for g in got]))
    assert np.abs(allp - np.sort(ref["peeq"])).max() < 1e-8 * ref["peeq"].max()


def _worker_emb(rank, world, port, out, level2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fcvm_hip import examples
        from fcvm_hip.dist import DistBackend, DistModel, HipLocalOps, Partition
        from fcvm_hip.loads import Loads, external_loads
        from fcvm_hip.solver import calc_disp
        torch.cuda.set_device(0)
        case = examples.embankment(refine=1)
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
        # "sparse": the aggregate level a mesh of more than 3000 aggregates gets (block-Jacobi Chebyshev sweeps on the replicated
        # block matrix, dense level of super-aggregates) forced onto this small mesh
        be.enable_multigrid(xyz_loc, **(dict(target=16, max_dense=12) if level2 == "sparse" else {}))
        model = DistModel(part, be, xyz_loc, case["material"], fix_loc, Loads(), external_loads)
        ctl = dict(case["ctl"], nstep=8)
        res = calc_disp(model, ctl, rtol=1e-11, maxit=4000)
        nshare = np.zeros(part.nn, int)
        for s, ix in part.shared.items():
            nshare[ix] += 1
        payload = dict(gnodes=part.gnodes, disp=res["displacements"], lout=res["lout"], un=res["un"], crip=res["crip"],
                       max_sharing=int(nshare.max()) + 1, pcg=res["stats"]["pcg_iterations"], solves=res["stats"]["solves"],
                       l2=be.mg.l2_info, lib=ops.c.precond_info()["level2"])
        got = [None] * world
        dist.all_gather_object(got, payload)
        if rank == 0:
            torch.save(got, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("level2", ["dense", "sparse"])
def test_three_ranks_unstructured_partition(tmp_path, level2):
    """Element-range partition of an UNSTRUCTURED mesh (the reference's Embankment model refined once,
    5,272 tet10) over three ranks: ragged interfaces, nodes held by all three ranks, distributed
    three-level multigrid with the REPLICATED aggregate level (blocks of Z^T A1 Z from k_galerkin_blocks on every rank, summed
    in the global block pattern; dense inverse or the sparse level under super-aggregates) -- against the single-GPU analysis."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r.pt")
    spawn_sharing_gpu(_worker_emb, (3, port, out, level2), 3)
    got = torch.load(out, weights_only=False)
    assert all(g["l2"] == got[0]["l2"] for g in got) and got[0]["l2"]["sparse"] == (level2 == "sparse"), [g["l2"] for g in got]
    assert ("sparse" in got[0]["lib"]) == (level2 == "sparse"), got[0]["lib"]
    print("three ranks (%s):" % level2, got[0]["l2"], got[0]["pcg"] / got[0]["solves"], "PCG iterations per solve")
    from fcvm_hip import examples
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    case = examples.embankment(refine=1)
    ref = run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(), dict(case["ctl"], nstep=8), rtol=1e-11)
    assert ref["peeq"].max() > 0
    assert max(g["max_sharing"] for g in got) >= 3
    rel = lambda a, b: np.abs(np.asarray(a) - b).max() / np.abs(b).max()  # noqa: E731
    for g in got:
        assert len(g["lout"]) == len(ref["lout"])
        assert rel(g["lout"], ref["lout"]) < 1e-6 and rel(g["un"], ref["un"]) < 1e-6
        assert np.array_equal(g["crip"], ref["crip"])
        d = ref["displacements"].reshape(-1, 3)[g["gnodes"] - 1].reshape(-1)
        assert rel(g["disp"], d) < 1e-6
    assert got[0]["pcg"] / got[0]["solves"] < 400


COLUMN_CTL = dict(sig_yield=240.0, grav_x=0.0, grav_y=0.0, grav_z=0.0, nstep=5, iterat_max=20, error_max=1e-4, relax=1.2,
                  scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25, Et_E=0.0, target_LF=10.0,
                  csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLY", maxImp="0.5", ev1="1.0", ev2="0.3")


def _column():
    from fcvm_hip import meshgen
    from fcvm_hip.loads import Loads
    el, xyz = meshgen.cube_tet10(10, 2, 3, 40.0, 6.0, 10.0)
    fix = {3 * int(n) + k: 0.0 for n in meshgen.nodes_on_plane(xyz, 0, 0.0) for k in range(3)}
    faces = meshgen.boundary_faces_on_plane(el, xyz, 0, 40.0)
    return el, xyz, (210000.0, 0.3, 7.8e-6), fix, Loads(ufaces=faces, uloads=np.tile([-5000.0, 0.0, 0.0], (len(faces), 1)))


def _worker_buckling(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fcvm_hip.dist import partition_model
        from fcvm_hip.solver import calc_disp
        torch.cuda.set_device(0)
        el, xyz, mat, fix, loads = _column()
        order = np.random.default_rng(5).permutation(len(el))   # ragged interfaces, the loaded face split over the ranks
        model = partition_model(el, xyz, mat, fix, loads, 0, order=order)
        res = calc_disp(model, COLUMN_CTL, rtol=1e-11, maxit=4000)
        payload = dict(gnodes=model.part.gnodes, disp=res["displacements"], lout=res["lout"], un=res["un"],
                       eigenval=res["eigenval"], nocoord=res["nocoord"], eigenvec=res["eigenvec"])
        got = [None] * world
        dist.all_gather_object(got, payload)
        if rank == 0:
            torch.save(got, out)
    finally:
        dist.destroy_process_group()


def test_gnly_buckling_preanalysis_on_a_partitioned_model(tmp_path):
    """The GNLY flow of fcVM.py:1199-1296 on an element partition: elastic solution, buckling factors by subspace iteration
    with distributed multigrid-PCG solves on K_b + sigma K_g, imperfection scaled by the GLOBAL extreme of the mixed buckling
    shape, restart on the imperfect geometry -- three ranks on GPU 0 against the single-GPU run of the same column."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r.pt")
    spawn_sharing_gpu(_worker_buckling, (3, port, out), 3)
    got = torch.load(out, weights_only=False)
    from fcvm_hip.solver import run_analysis
    el, xyz, mat, fix, loads = _column()
    ref = run_analysis(el, xyz, mat, fix, loads, COLUMN_CTL, rtol=1e-11)
    rel = lambda a, b: np.abs(np.asarray(a) - b).max() / np.abs(b).max()  # noqa: E731
    assert np.abs(ref["nocoord"] - xyz).max() == pytest.approx(0.5, rel=1e-9)
    for g in got:
        assert np.allclose(g["eigenval"], ref["eigenval"], rtol=1e-8)
        assert rel(g["nocoord"], ref["nocoord"][g["gnodes"] - 1]) < 1e-8          # the same imperfection on every rank
        assert len(g["lout"]) == len(ref["lout"])
        assert rel(g["lout"], ref["lout"]) < 1e-6 and rel(g["un"], ref["un"]) < 1e-6
        assert rel(g["disp"], ref["displacements"].reshape(-1, 3)[g["gnodes"] - 1].reshape(-1)) < 1e-6
