"""HIP path (through the C-ABI of libfcvm_hip.so) against the CPU oracle and the reference's
golden result files.  FP64 throughout; tolerances are relative to the field's max magnitude (and, where rel_each is used, entry by entry):
1e-11 for single kernels (only summation order differs), 1e-6 (north_star) for whole analyses."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import load_golden, small_case

pytestmark = pytest.mark.gpu

KTOL = 1e-11


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def rel_each(a, b, floor=1e-3):
    """largest ELEMENT-WISE relative difference over the entries of b above floor * max |b| (the round-2 review noted that `rel`
    measures against the field maximum only)"""
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    m = np.abs(b) > floor * np.abs(b).max()
    return float((np.abs(a - b)[m] / np.abs(b)[m]).max()) if m.any() else 0.0


def make_model(case, loads=True):
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    ld = Loads(ufaces=case["ufaces"], uloads=case["uloads"]) if loads else Loads()
    return GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], ld)


def oracle_full_matrix(orc, case, grav, **kw):
    stm, row, col, glv, modf, V, *_ls, ne, nn, x = orc.calcGSM(
        case["elNodes"], case["nocoord"], case["material"], case["fix"], grav,
        orc.Loads(ufaces=case["ufaces"], uloads=case["uloads"]), **kw)
    n = 3 * len(case["nocoord"])
    low = sp.coo_matrix((stm, (row, col)), shape=(n, n)).tocsr()
    full = low + sp.tril(low, -1).T
    return full.tocsr(), glv, modf, V, x


def gpu_matrix(model):
    r, c, v = model.ctx.matrix_coo()
    n = model.ndof
    return sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()


@pytest.mark.parametrize("n", [1, 3])
def test_assemble_matches_calcGSM(orc, n):
    case = small_case(n)
    grav = (0.3, -0.2, -9.81)
    Ko, glv_o, modf_o, V_o, x_o = oracle_full_matrix(orc, case, grav)
    m = make_model(case)
    glv, modf, V, xgp = m.calc_gsm(grav)
    Kg = gpu_matrix(m)
    assert abs(Kg - Ko).max() <= KTOL * abs(Ko).max()
    assert rel(glv.get(), glv_o) < KTOL
    assert rel(modf.get(), modf_o) < KTOL
    assert rel(xgp.get(), x_o.reshape(-1)) < KTOL
    assert abs(V - V_o) < 1e-11 * abs(V_o)
    # fixed dofs: unit diagonal per connected element, nothing else in row / column
    d = Kg.diagonal()
    noce = orc.node_element_count(case["elNodes"], len(case["nocoord"]))
    for dof in case["fix"]:
        assert d[dof] == noce[dof // 3]
        assert Kg[dof].nnz == 1 or abs(Kg[dof]).sum() == d[dof]


@pytest.mark.parametrize("host_map", [False, True])
def test_element_map_built_on_the_device_equals_the_host_builders(monkeypatch, host_map, embankment):
    """fcvm_set_mesh fills the element -> block-location table of the assembly kernels with k_build_emap; every entry must be what
    fcvm_pattern.cpp's rule gives (and the uploaded host table under FCVM_EMAP_HOST=1 passes the same check): the structured
    cube (caller's row order, odd slice lengths), the unstructured Embankment mesh (Morton row order) and a shuffled mesh
    whose slices mix long and short rows."""
    from fcvm_hip import meshgen
    if host_map:
        monkeypatch.setenv("FCVM_EMAP_HOST", "1")
    from fcvm_hip import examples
    g, _ = embankment
    cases = [small_case(3), meshgen.cube_case(9, l=10.0, traction=(0.0, 0.0, -1.0)), dict(elNodes=g["elNodes"], nocoord=g["nocoord"]),
             examples.embankment(refine=1)]
    rng = np.random.default_rng(5)
    c9 = meshgen.cube_case(9, l=10.0, traction=(0.0, 0.0, -1.0))
    perm = rng.permutation(len(c9["nocoord"]))            # node i of the shuffled mesh = node perm[i] of the cube
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    el = inv[np.asarray(c9["elNodes"]) - 1] + 1
    cases.append(dict(elNodes=el[rng.permutation(len(el))], nocoord=np.asarray(c9["nocoord"])[perm]))
    from fcvm_hip.device import Context
    for case in cases:
        ctx = Context(0)
        ctx.set_mesh(np.asarray(case["elNodes"]), np.asarray(case["nocoord"], dtype=np.float64))
        n, bad = ctx.check_element_map()
        assert n == 100 * len(case["elNodes"]) and bad == 0
        ctx.close()


def _random_state(case, rng, scale):
    nn, ne = len(case["nocoord"]), len(case["elNodes"])
    du = rng.standard_normal(3 * nn) * scale
    disp = rng.standard_normal(3 * nn) * scale
    sig = rng.standard_normal(24 * ne) * 50.0
    sy = np.full(4 * ne, 200.0) + rng.random(4 * ne) * 50.0
    return du, disp, sig, sy


@pytest.mark.parametrize("LD", [False, True])
def test_update_stress_load_matches(orc, gpu_ctx, LD):
    case = small_case(3)
    rng = np.random.default_rng(5)
    du, disp, sig, sy = _random_state(case, rng, 2e-3)
    ne, nn = len(case["elNodes"]), len(case["nocoord"])
    s_new, s_test, q = np.zeros(24 * ne), np.zeros(24 * ne), np.zeros(3 * nn)
    pg = np.zeros(4 * ne, np.uint8)
    orc.update_stress_load(case["elNodes"], case["nocoord"], case["material"], sy, disp, du, sig, s_new, s_test, q,
                           0.05, LD, pg)
    assert 0 < pg.sum() < len(pg), "test state must mix elastic and plastic Gauss points"
    m = make_model(case)
    c = m.ctx
    d_new, d_test, d_q, d_pg = c.empty(24 * ne), c.empty(24 * ne), c.empty(3 * nn), c.empty(4 * ne, np.uint8)
    c.update_stress_load(c.array(sy), c.array(disp), c.array(du), c.array(sig), d_new, d_test, d_q, 0.05, LD, d_pg)
    assert np.array_equal(d_pg.get(), pg)
    assert rel(d_test.get(), s_test) < KTOL
    assert rel(d_new.get(), s_new) < KTOL
    assert rel(d_q.get(), q) < KTOL
    assert rel_each(d_test.get(), s_test) < 1e-9 and rel_each(d_new.get(), s_new) < 1e-9 and rel_each(d_q.get(), q) < 1e-8


def test_tangent_assemble_matches_calcTSM(orc):
    case = small_case(3)
    rng = np.random.default_rng(7)
    du, disp, sig, sy = _random_state(case, rng, 1e-3)
    ne = len(case["elNodes"])
    pgp = (rng.random(4 * ne) < 0.4).astype(np.uint8)
    grav = (0.0, 0.0, -9.81)
    Ko, glv_o, modf_o, _, _ = oracle_full_matrix(orc, case, grav, disp=disp, sig=sig, pgp=pgp, Et_E=0.1)
    m = make_model(case)
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm(grav, disp=c.array(disp), sig=c.array(sig), pgp=c.array(pgp, np.uint8), Et_E=0.1)
    Kg = gpu_matrix(m)
    assert abs(Kg - Ko).max() <= KTOL * abs(Ko).max()
    assert rel(modf.get(), modf_o) < KTOL
    # pressure-free model: glv = uniform face load + gravity on the displaced geometry
    assert rel(glv.get(), glv_o) < KTOL


@pytest.mark.parametrize("mesh", ["cube", "embankment"])
def test_assembly_straight_into_the_symmetric_window_form(orc, monkeypatch, mesh):
    """Round 4: on meshes that run the symmetric window form, calcGSM / calcTSM (fcVM.py:729-800, 940-1050) leave the assembly
    kernel's LDS straight in the SpMV's storage (FP64 values of the stored blocks, their 16-bit copy, the diagonal blocks) --
    no sliced-ELL array, no copy kernel.  The matrix read back from that storage (stored blocks + transposes) must be the
    oracle's, elastic and tangent form; load vectors, Gauss point coordinates and volume as before; and everything the solver
    sees -- FP64 product, 16-bit product, cycle, solution -- must be BITWISE what the sliced-ELL + copy path of rounds 2-3
    gives (same sums in the same order, only the route of the values differs).  Cube: structured, prescribed displacements,
    corner rows cut into segments; Embankment refined once: unstructured, gravity."""
    from fcvm_hip import examples
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    if mesh == "cube":
        case = small_case(7)
        grav = (0.3, -0.2, -9.81)
    else:
        case = examples.embankment(refine=1)
        case["material"] = np.asarray(case["material"], dtype=np.float64)
        grav = (0.0, 0.0, -10.0)
    if "ufaces" not in case:
        case["ufaces"], case["uloads"] = np.zeros((0, 6), np.int64), np.zeros((0, 3))
    rng = np.random.default_rng(3)
    du, disp, sig, sy = _random_state(case, rng, 1e-3)
    ne, n = len(case["elNodes"]), 3 * len(case["nocoord"])
    pgp = (rng.random(4 * ne) < 0.4).astype(np.uint8)
    x = rng.standard_normal(n)
    res = {}
    for route in ("direct", "copy"):
        monkeypatch.setenv("FCVM_ASSEMBLE_SYM", "1" if route == "direct" else "0")
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]),
                     ctx=Context(0))
        c = m.ctx
        out = {}
        for form in ("elastic", "tangent"):
            kw = {} if form == "elastic" else dict(disp=c.array(disp), sig=c.array(sig), pgp=c.array(pgp, np.uint8), Et_E=0.1)
            glv, modf, V, xgp = m.calc_gsm(grav, **kw)
            info = c.spmv_sym_info()
            assert info["built"] and info["in_use"] and info["assembly_writes_it"] == (route == "direct")
            assert info["assembled_directly"] == (route == "direct")
            xd = c.array(m.fixdof_h * x)
            f = c.mul(m.fixdof, glv)
            u, it = c.solve(f, rtol=1e-11)
            out[form] = dict(K=gpu_matrix(m), glv=glv.get(), modf=modf.get(), V=V, xgp=xgp.get(), y=c.spmv(xd, c.empty(n)).get(),
                             ylo=c.spmv_lo(xd, c.empty(n)).get(), z=c.mg_apply(xd, c.empty(n)).get(), u=u.get(), it=it["iterations"],
                             dblk=c.get_diag_blocks().get())
        res[route] = out
        c.close()
    for form in ("elastic", "tangent"):
        kw = {} if form == "elastic" else dict(disp=disp, sig=sig, pgp=pgp, Et_E=0.1)
        Ko, glv_o, modf_o, V_o, x_o = oracle_full_matrix(orc, case, grav, **kw)
        d, cp = res["direct"][form], res["copy"][form]
        assert abs(d["K"] - Ko).max() <= KTOL * abs(Ko).max()
        assert abs(d["K"] - d["K"].T).max() <= 1e-15 * abs(Ko).max()    # (pairs inside a group: one stored block serves both halves)
        assert rel(d["glv"], glv_o) < KTOL and rel(d["modf"], modf_o) < KTOL and rel(d["xgp"], x_o.reshape(-1)) < KTOL
        assert abs(d["V"] - V_o) < 1e-11 * abs(V_o)
        xm = x.copy()
        xm[list(case["fix"])] = 0.0
        assert rel(d["y"], Ko @ xm) < 1e-12
        diag = np.stack([Ko[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray().reshape(9) for i in range(n // 3)]).reshape(-1)
        assert rel(d["dblk"], diag) < KTOL
        for k in ("glv", "xgp", "y", "ylo", "z", "dblk"):
            assert np.array_equal(d[k], cp[k]), (form, k)
        # modf = - sum over the row's blocks of K_ab u_b(prescribed): the lanes of a row sum their own columns first (eight lanes
        # per row on this route, four on the copy route), so its rounding -- and through the right-hand side the solution's -- differs
        assert rel(d["modf"], cp["modf"]) < 1e-14 and rel(d["u"], cp["u"]) < 1e-9
        assert abs(d["V"] - cp["V"]) <= 1e-13 * abs(V_o) and abs(d["it"] - cp["it"]) <= 1      # (volume partials are summed per work item)
        assert abs(d["K"] - cp["K"]).max() <= 1e-15 * abs(Ko).max()    # (the copy route reads the sliced-ELL array: both halves stored)


def test_update_peeq_csr_matches(orc, gpu_ctx):
    case = small_case(2)
    rng = np.random.default_rng(11)
    ne = len(case["elNodes"])
    mat = case["material"]
    st, sn = rng.standard_normal(24 * ne) * 150, rng.standard_normal(24 * ne) * 100
    arrs = dict(sy=np.full(4 * ne, 235.0), peeq=rng.random(4 * ne) * 1e-3, csr=rng.random(4 * ne) * 1e-2)
    host = {k: v.copy() for k, v in arrs.items()}
    tri, pre, svm, ecr = (np.zeros(4 * ne) for _ in range(4))
    orc.update_PEEQ_CSR(ne, mat, st, sn, host["sy"], 0.25, host["peeq"], host["csr"], tri, pre, svm, ecr, 0.02)
    m = make_model(case)
    c = m.ctx
    d = {k: c.array(v) for k, v in arrs.items()}
    dt, dp, ds, de = (c.zeros(4 * ne) for _ in range(4))
    o = c.update_peeq_csr(c.array(st), c.array(sn), d["sy"], 0.25, d["peeq"], d["csr"], dt, dp, ds, de, 0.02)
    for k in arrs:
        assert rel(d[k].get(), host[k]) < 1e-13, k
    assert rel(dt.get(), tri) < 1e-13 and rel(dp.get(), pre) < 1e-13 and rel(ds.get(), svm) < 1e-13
    assert rel(de.get(), ecr) < 1e-12
    ml = int(np.argmax(host["csr"]))
    assert o["maxloc"] == ml
    assert o["csr_max"] == pytest.approx(host["csr"][ml], rel=1e-13)
    assert o["peeq_max"] == pytest.approx(host["peeq"].max(), rel=1e-13)
    assert o["pressure"] == pytest.approx(pre[ml], rel=1e-13)
    assert o["sigmises"] == pytest.approx(svm[ml], rel=1e-13)


@pytest.mark.parametrize("averaged", [False, True])
def test_map_stresses_and_principal(orc, gpu_ctx, averaged):
    case = small_case(3)
    rng = np.random.default_rng(3)
    ne, nn = len(case["elNodes"]), len(case["nocoord"])
    sig, peeq = rng.standard_normal(24 * ne) * 100, rng.random(4 * ne)
    svm, csr = rng.random(4 * ne) * 200, rng.random(4 * ne)
    noce = orc.node_element_count(case["elNodes"], nn)
    ref = orc.mapStresses(averaged, case["elNodes"], case["nocoord"], sig, peeq, svm, csr, noce, 235.0)
    m = make_model(case)
    c = m.ctx
    out = c.map_stresses(averaged, c.array(sig), c.array(peeq), c.array(svm), c.array(csr), 235.0)
    assert rel(out[0].get(), ref[0].reshape(-1)) < 1e-12
    for k in range(1, 5):
        assert rel(out[k].get(), ref[k]) < 1e-12, k
    ps, pv = c.principal_stress(out[0])
    s1, s2, s3, v1, v2, v3 = orc.calculate_principal_stress(ref[0])
    ps = ps.get().reshape(-1, 3)
    pv = pv.get().reshape(-1, 3, 3)
    scale = np.abs(ref[0]).max()
    assert np.abs(ps - np.stack([s1, s2, s3], 1)).max() < 1e-12 * scale
    for k, v in enumerate((v1, v2, v3)):
        d = np.minimum(np.abs(pv[:, k] - v).max(1), np.abs(pv[:, k] + v).max(1))
        assert d.max() < 1e-9 * scale


def test_pcg_solve_matches_direct(orc):
    case = small_case(4)
    Ko, glv_o, modf_o, _, _ = oracle_full_matrix(orc, case, (0, 0, -9.81))
    m = make_model(case)
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm((0, 0, -9.81))
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    u, info = c.solve(f, rtol=1e-12)
    fixdof = m.fixdof_h
    u_ref = spla.spsolve(Ko.tocsc(), fixdof * glv_o + modf_o)
    assert info["rel_residual"] <= 1e-12
    assert rel(u.get(), u_ref) < 1e-9
    for dof, val in case["fix"].items():
        assert u.get()[dof] == pytest.approx(val, abs=1e-12)
    # SpMV and preconditioner on their own
    x = np.random.default_rng(1).standard_normal(m.ndof)
    y = c.spmv(c.array(x), c.empty(m.ndof)).get()
    assert rel(y, Ko @ x) < 1e-13
    z = c.precond(c.array(x), c.empty(m.ndof)).get()
    blocks = [np.linalg.inv(Ko[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray()) @ x[3 * i:3 * i + 3] for i in range(m.nn)]
    assert rel(z, np.concatenate(blocks)) < 1e-10


def _graded_shell(n_t, n_a, ratio, ang=0.7):
    """A sector of a thick spherical shell, radii 1 .. ratio in geometric progression (n_t cells), n_a x n_a cells across a patch of `ang`
    radians: cells that shrink ISOTROPICALLY towards the inner face -- element size, and with it the stiffness entries, span 1 : ratio
    (a cube lattice graded per axis makes slivers whose entries span the SQUARE of the size ratio).  Outer face clamped, traction on the
    inner face; straight-sided elements (mid-side nodes put back at the edge midpoints by merge_blocks)."""
    from fcvm_hip import meshgen
    el, q = meshgen.cube_tet10(n_t, n_a, n_a, 1.0, 1.0, 1.0)
    d = np.stack([np.tan((q[:, 1] - 0.5) * ang), np.tan((q[:, 2] - 0.5) * ang), np.ones(len(q))], 1)
    xyz = (ratio ** q[:, 0] / np.linalg.norm(d, axis=1))[:, None] * d
    el, xyz = meshgen.merge_blocks([(el, xyz)], lattices=[(2 * n_t + 1, 2 * n_a + 1, 2 * n_a + 1)])
    fix = {3 * int(nd) + k: 0.0 for nd in np.where(np.abs(q[:, 0] - 1.0) < 1e-12)[0] for k in range(3)}
    on = np.abs(q[:, 0]) < 1e-12
    faces = []
    for f in meshgen.TET_FACES:
        nd = el[:, list(f)]
        faces.append(nd[on[nd[:, 0] - 1] & on[nd[:, 1] - 1] & on[nd[:, 2] - 1]])
    faces = np.concatenate(faces)
    return dict(elNodes=el, nocoord=xyz, fix=fix, ufaces=faces, uloads=np.tile([1.0, 0.5, 8.0], (len(faces), 1)),
                material=np.array([210000.0, 0.3, 7.8e-6]))


@pytest.mark.parametrize("ratio,n_t,declined", [(1.0e3, 40, False), (1.0e5, 66, True)], ids=["graded-1e3", "graded-1e5"])
def test_fixed_point_symmetric_form_on_graded_meshes(ratio, n_t, declined):
    """The symmetric window form adds its transposed products in 64-bit fixed point with ONE unit per group taken from the largest
    matrix entry (fcvm_spmv_win.hip.h): on a graded mesh the products in the region of smallest entries lose relative precision as
    max / min of the nodal diagonal entries.  Round 5: fcvm_assemble measures that ratio (k_diag_grading) and declines the form
    beyond FCVM_SYM_GRADING_MAX = 1e4.  A shell sector graded 1 : 1e3 in element size (nodal diagonal entries 1 : 7.6e3) keeps the form
    and still solves to 1e-11 with a TRUE residual at that level; graded 1 : 1e5 (7.6e5) the context falls back to the full window form
    (fcvm_spmv_sym_info says so) and solves as well."""
    case = _graded_shell(n_t, 4, ratio)
    m = make_model(case)
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm((0, 0, -9.81))
    si, wi = c.spmv_sym_info(), c.spmv_info()
    assert wi["window_format"] and si["built"], (wi, si)
    assert si["declined_graded"] == declined and si["in_use"] == (not declined), si
    assert (5e3 < si["grading"] < 1e4) if not declined else si["grading"] > 1e5, si
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    u, info = c.solve(f, rtol=1e-11, maxit=5000)
    K = gpu_matrix(m).tocsr()       # (read back from the form that is in charge)
    fh, uh = f.get(), u.get()
    true_res = np.linalg.norm(fh - K @ uh) / np.linalg.norm(fh)
    assert info["rel_residual"] <= 1e-11 and true_res < 4e-11, (info, true_res)
    # the product itself, entry by entry against the host's (rows of the smallest elements included: relative to each ROW's scale)
    x = np.random.default_rng(2).standard_normal(m.ndof)
    y, yh = c.spmv(c.array(x), c.empty(m.ndof)).get(), K @ x
    rowscale = np.asarray(abs(K).sum(axis=1)).reshape(-1)
    assert (np.abs(y - yh) / rowscale).max() < (1e-12 if not declined else 1e-14)
    u_ref = spla.spsolve(K.tocsc(), fh)
    assert rel(uh, u_ref) < 1e-6          # (conditioning of a 1 : 1e3 .. 1e5 graded mesh, not the solver, sets this)


@pytest.mark.parametrize("mode", ["bj", "mg"])
def test_both_preconditioners_solve_the_embankment_system(embankment, mode):
    """unstructured Gmsh mesh, nu = 0.45: block-Jacobi and p-multigrid PCG reach the same solution"""
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    g, _ = embankment
    fix = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    m = GpuModel(g["elNodes"], g["nocoord"], g["material"], fix, Loads())
    c = m.ctx
    c.set_precond(mode)
    glv, modf, _, _ = m.calc_gsm((0.0, 0.0, -10.0))
    f = c.mul(m.fixdof, glv)
    u, info = c.solve(f, rtol=1e-12)
    r = c.axpby(1.0, f, -1.0, c.spmv(u, c.empty(m.ndof)))
    assert c.norm(r) <= 5e-12 * c.norm(f)
    K = gpu_matrix(m)
    u_ref = spla.spsolve(K.tocsc(), f.get())
    assert rel(u.get(), u_ref) < 1e-8
    if mode == "mg":
        assert info["iterations"] < 200 and c.precond_info()["coarse_nodes"] > 0


def test_16bit_smoother_cycle_is_a_symmetric_preconditioner_and_solves(monkeypatch):
    """The multiplicative V(1,1) cycle whose two fine-level smoother SpMVs stream a 16-bit fixed-point copy of the matrix (cycle form 2,
    k_spmv_win_lo): as an operator z = B r it must be symmetric and positive definite (PCG's requirement; checked with random
    vectors through fcvm_mg_apply), the PCG built on it must reach the solution of the FP64 matrix (true residual from the
    FP64 SpMV, sparse direct solve of the matrix read back), in fewer iterations than the additive cycle, and with groups
    of 4 waves as well (windows forced small).  Embankment refined once (unstructured, nu = 0.45) and a cube with split rows."""
    from fcvm_hip import examples, meshgen
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    emb = examples.embankment(refine=1)
    cube = meshgen.cube_case(12, l=10.0, traction=(0.0, 0.0, 1.0))
    cube["material"] = (210000.0, 0.3, 7.8e-6)
    rng = np.random.default_rng(11)
    for case, grav in ((emb, (0.0, 0.0, -10.0)), (cube, (0.0, 0.0, 0.0))):
        its = {}
        for form, small in ((1, False), (2, False), (2, True)):
            if small:
                monkeypatch.setenv("FCVM_WIN_MAX", str(wmax - 1))
            ctx = Context(0)
            ctx.set_mg_cycle(form)
            m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(), ctx=ctx)
            monkeypatch.delenv("FCVM_WIN_MAX", raising=False)
            c = m.ctx
            wmax = c.spmv_info()["max_window"]
            assert c.spmv_info()["waves_per_group"] == (4 if small else 8)
            glv, modf, _, _ = m.calc_gsm(grav, want_x=False)
            f = c.mul(m.fixdof, glv) if np.abs(glv.get()).max() > 0 else c.array(m.fixdof_h * rng.standard_normal(m.ndof))
            if form == 2:   # B symmetric positive definite on the free dofs
                a, b = (c.array(m.fixdof_h * rng.standard_normal(m.ndof)) for _ in range(2))
                Ba, Bb = c.mg_apply(a, c.empty(m.ndof)), c.mg_apply(b, c.empty(m.ndof))
                ab, ba, aa, bb = c.dot(b, Ba), c.dot(a, Bb), c.dot(a, Ba), c.dot(b, Bb)
                assert abs(ab - ba) <= 1e-6 * np.sqrt(aa * bb), (ab, ba)    # rounding of K_ij / K_ji may differ in the last bit
                assert aa > 0 and bb > 0
            u, info = c.solve(f, rtol=1e-12)
            r = c.axpby(1.0, f, -1.0, c.spmv(u, c.empty(m.ndof)))
            assert c.norm(r) <= 5e-12 * c.norm(f)
            if not small:
                K = gpu_matrix(m)
                assert rel(u.get(), spla.spsolve(K.tocsc(), f.get())) < 1e-8
            its[(form, small)] = info["iterations"]
            c.close()
        assert its[(2, False)] < 0.8 * its[(1, False)], its
        assert abs(its[(2, True)] - its[(2, False)]) <= 2, its


def test_two_right_hand_sides_in_lockstep_match_single_solves_bitwise():
    """fcvm_solve2 (the GNLY pair K ue = f, K due = r of fcVM.py:1351-1401): every matrix pass serves both right-hand sides
    (k_spmv_win2, k_spmv_win_lo2), the rest of each PCG runs on its own state.  Each result must be bitwise what fcvm_solve
    gives for that right-hand side alone -- also when the two converge at different iterations (different tolerances: the
    finished one idles) and with split rows (cube corner nodes); and it must solve the system (true residual)."""
    from fcvm_hip import examples, meshgen
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    emb = examples.embankment(refine=2)
    cube = meshgen.cube_case(12, l=10.0, traction=(0.0, 0.0, 1.0))
    cube["material"] = (210000.0, 0.3, 7.8e-6)
    rng = np.random.default_rng(5)
    for case, grav in ((emb, (0.0, 0.0, -10.0)), (cube, (0.0, 0.0, 0.0))):
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                     Loads(ufaces=case["ufaces"], uloads=case["uloads"]) if "ufaces" in case else Loads())
        c = m.ctx
        glv, modf, _, _ = m.calc_gsm(grav, want_x=False)
        assert c.spmv_info()["window_format"] and c.precond_info()["cycle_form"] == 2
        fa = c.mul(m.fixdof, glv)
        fb = c.array(m.fixdof_h * rng.standard_normal(m.ndof))
        e = lambda: c.empty(m.ndof)  # noqa: E731
        ya, yb = c.spmv2(fa, fb, e(), e())                      # the shared FP64 pass on its own ...
        assert np.array_equal(ya.get(), c.spmv(fa, e()).get()) and np.array_equal(yb.get(), c.spmv(fb, e()).get())
        za, zb = c.mg_apply2(fa, fb, e(), e())                  # ... and the cycle with its two shared 16-bit passes
        assert np.array_equal(za.get(), c.mg_apply(fa, e()).get()) and np.array_equal(zb.get(), c.mg_apply(fb, e()).get())
        for rtol in ((1e-11, 1e-11), (1e-12, 1e-6), (1e-5, 1e-11)):
            ua, ia = c.solve(fa, rtol=rtol[0])
            ub, ib = c.solve(fb, rtol=rtol[1])
            va, vb, ja, jb = c.solve2(fa, fb, rtol=rtol)
            assert (ja["iterations"], jb["iterations"]) == (ia["iterations"], ib["iterations"]), (ia, ib, ja, jb)
            assert np.array_equal(va.get(), ua.get()) and np.array_equal(vb.get(), ub.get())
            for f, v, t in ((fa, va, rtol[0]), (fb, vb, rtol[1])):
                r = c.axpby(1.0, f, -1.0, c.spmv(v, c.empty(m.ndof)))
                assert c.norm(r) <= 5.0 * t * c.norm(f)
        c.close()


def test_window_format_spmv_and_pcg_match_sliced_ell(monkeypatch):
    """The SpMV's second storage of the fine matrix (window format: Morton groups, x of a group's column nodes in LDS,
    16-bit window columns, prefix storage without padding, long rows cut into segments) against the sliced-ELL kernel and
    against scipy on the matrix read back from the device -- unstructured mesh (Embankment refined twice) and a cube whose
    corner-node rows (65 blocks) are split in three; then the PCG solve, whose p.Ap comes from the kernel's own partials."""
    from fcvm_hip import examples, meshgen
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    emb = examples.embankment(refine=2)
    cube = meshgen.cube_case(12, l=10.0, traction=(0.0, 0.0, 1.0))
    cube["material"] = (210000.0, 0.3, 7.8e-6)
    for case, grav in ((emb, (0.0, 0.0, -10.0)), (cube, (0.0, 0.0, 0.0))):
        models = {}
        for win in ("1", "0", "small"):     # "small": a window limit that only groups of 4 waves (256 rows) meet
            monkeypatch.setenv("FCVM_SPMV_WIN", "0" if win == "0" else "1")
            if win == "small":
                monkeypatch.setenv("FCVM_WIN_MAX", str(models["1"].ctx.spmv_info()["max_window"] - 1))
            ctx = Context(0)
            ctx.set_mg_cycle(1)     # one cycle form for all three storages (form 2 needs the window format: its own test)
            m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(), ctx=ctx)
            assert m.ctx.spmv_info()["waves_per_group"] == {"1": 8, "0": 0, "small": 4}[win]
            models[win] = m
        monkeypatch.delenv("FCVM_SPMV_WIN")
        monkeypatch.delenv("FCVM_WIN_MAX")
        mw, ms = models["1"], models["0"]
        info = mw.ctx.spmv_info()
        assert mw.ctx.pattern_info()["nnz_blocks"] <= info["blocks"] < mw.ctx.pattern_info()["padded_slots"] and 0 < info["max_window"] <= 3200
        x = np.random.default_rng(3).standard_normal(mw.ndof)
        out = {}
        for k, m in models.items():
            glv, modf, _, _ = m.calc_gsm(grav, want_x=False)
            out[k] = (m.ctx.spmv(m.ctx.array(x), m.ctx.empty(m.ndof)).get(), glv, modf)
        K = gpu_matrix(ms)
        assert rel(out["0"][0], K @ x) < 1e-13 and rel(out["1"][0], K @ x) < 1e-13
        assert rel(out["1"][0], out["0"][0]) < 1e-14 and rel(out["small"][0], out["0"][0]) < 1e-14
        sols = {}
        for k, m in models.items():
            c = m.ctx
            f = c.mul(m.fixdof, out[k][1]) if np.abs(out[k][1].get()).max() > 0 else c.array(m.fixdof_h * x)
            u, it = c.solve(f, rtol=1e-12)
            r = c.axpby(1.0, f, -1.0, c.spmv(u, c.empty(m.ndof)))
            assert c.norm(r) <= 5e-12 * c.norm(f)
            sols[k] = (u.get(), it["iterations"])
        assert rel(sols["1"][0], sols["0"][0]) < 1e-9 and rel(sols["small"][0], sols["0"][0]) < 1e-9
        assert abs(sols["1"][1] - sols["0"][1]) <= 2          # same Krylov iteration, different summation order
        for m in models.values():
            m.ctx.close()


def test_sparse_aggregate_level_solves_like_the_dense_one(monkeypatch):
    """Meshes with more aggregates than a dense inverse carries keep A2 = Z^T A1 Z sparse (6 x 6 blocks, block-Jacobi Chebyshev
    smoothing) under a dense level of super-aggregates.  Forced here on the Embankment model refined twice with a limit of 40
    aggregates: the same solution as with the dense aggregate level and a comparable iteration count."""
    from fcvm_hip import examples
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    case = examples.embankment(refine=2)
    res = {}
    for name, env in (("dense", {}), ("sparse", {"FCVM_MG_MAX_AGGREGATES": "40"}),
                      ("sparse, symmetric product", {"FCVM_MG_MAX_AGGREGATES": "40", "FCVM_L2S_SYMV_MIN": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(), ctx=Context(0))
        c = m.ctx
        glv, modf, _, _ = m.calc_gsm((0.0, 0.0, -10.0), want_x=False)
        pi = c.precond_info()
        f = c.mul(m.fixdof, glv)
        u, info = c.solve(f, rtol=1e-11)
        r = c.axpby(1.0, f, -1.0, c.spmv(u, c.empty(m.ndof)))
        assert c.norm(r) <= 5e-11 * c.norm(f)
        res[name] = (u.get(), info["iterations"], pi)
        c.close()
        for k in env:
            monkeypatch.delenv(k)
    assert not res["dense"][2]["l2_sparse"] and res["dense"][2]["l2_aggregates"] > 40
    sp = res["sparse"][2]
    assert sp["l2_sparse"] and sp["l2_aggregates"] == res["dense"][2]["l2_aggregates"] and 0 < sp["l3_super_aggregates"] <= 40
    assert 0 < sp["l3_unknowns"] <= 6 * sp["l3_super_aggregates"]
    assert rel(res["sparse"][0], res["dense"][0]) < 1e-8
    assert res["sparse"][1] <= 1.15 * res["dense"][1] + 5, (res["sparse"][1], res["dense"][1])
    # the inverse of the super-aggregate level applied from its lower triangle (what levels of 2048 unknowns and more get): the same
    # preconditioner up to the order of the sums
    sy = res["sparse, symmetric product"]
    assert sy[2]["l2_sparse"] and sy[2]["l3_unknowns"] == sp["l3_unknowns"]
    assert rel(sy[0], res["dense"][0]) < 1e-8 and abs(sy[1] - res["sparse"][1]) <= 2, (sy[1], res["sparse"][1])


@pytest.mark.parametrize("mesh", ["plate", "embankment", "cube"])
def test_symmetric_window_form_spmv_equals_the_full_form(monkeypatch, mesh):
    """The PCG's FP64 SpMV stores one block of every pair K_ij, K_ji inside a group and adds the transposed product through LDS in
    64-bit fixed point (k_spmv_win_sym).  Against the full window form (FCVM_WIN_SYM=0): same product to FP64 rounding for random,
    tiny, huge and zero vectors, bitwise the same on a rerun (integer sums do not depend on the arrival order), the same p.Kp
    and PCG solution; meshes with split rows (corner nodes of tet10 meshes: 65 blocks) and unstructured numbering."""
    from fcvm_hip import examples, meshgen
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    if mesh == "plate":
        case = examples.plate_with_hole(4.0)
        loads = Loads(ufaces=case["ufaces"], uloads=case["uloads"])
    elif mesh == "embankment":
        case = examples.embankment(refine=2)
        loads = Loads()
    else:
        case = meshgen.cube_case(14, l=50.0, traction=(0.0, 0.0, -1.0))
        case["material"] = (210000.0, 0.3, 7.8e-6)
        loads = Loads(ufaces=case["ufaces"], uloads=case["uloads"])
    rng = np.random.default_rng(17)
    res = {}
    xs = None
    for form in ("full", "sym"):
        if form == "full":
            monkeypatch.setenv("FCVM_WIN_SYM", "0")
        else:
            monkeypatch.delenv("FCVM_WIN_SYM", raising=False)
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], loads, ctx=Context(0))
        c = m.ctx
        glv, modf, _, _ = m.calc_gsm((0.0, 0.0, -9.81), want_x=False)
        assert c.spmv_info()["window_format"]
        if xs is None:
            x0 = rng.standard_normal(m.ndof)
            xs = [x0, 1e-200 * x0, 1e150 * x0, np.zeros(m.ndof), np.where(np.arange(m.ndof) % 97 == 0, 1.0, 0.0)]
        ys = [c.spmv(c.array(x), c.empty(m.ndof)).get() for x in xs]
        ys2 = [c.spmv(c.array(x), c.empty(m.ndof)).get() for x in xs]
        # the product with the 16-bit copy (FP32 arithmetic) on the same two forms
        from fcvm_hip._lib import call
        los = []
        for x in (xs[0], xs[4]):
            dx, dy = c.array(x), c.empty(m.ndof)
            call("fcvm_spmv_lo", c.h, dx.ptr, dy.ptr)
            los.append(dy.get())
        assert c.spmv_sym_info()["in_use"] == (form == "sym")
        f = c.mul(m.fixdof, glv)
        c.axpby(1.0, f, 1.0, modf, out=f)
        u, info = c.solve(f, rtol=1e-10)
        res[form] = (ys, ys2, u.get(), info["iterations"], los)
        c.close()
    for a, b in zip(res["sym"][0], res["sym"][1]):
        assert np.array_equal(a, b)
    for x, a, b in zip(xs, res["sym"][0], res["full"][0]):
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= 1e-13 * scale if scale > 0 else not a.any()
    for a, b in zip(res["sym"][4], res["full"][4]):
        assert np.abs(a - b).max() <= 5e-6 * np.abs(b).max()     # FP32 products in another order; both ~1e-5 off the FP64 product
    for a, b in zip(res["sym"][4], (res["full"][0][0], res["full"][0][4])):
        assert np.abs(a - b).max() <= 1e-3 * np.abs(b).max()
    assert abs(res["sym"][3] - res["full"][3]) <= 1
    assert rel(res["sym"][2], res["full"][2]) < 1e-8


def test_aggregate_level_product_from_the_lower_triangle_equals_the_row_kernel(monkeypatch):
    """e2 = A2^-1 r2 is applied from the lower triangle of the FP32 inverse (k_symv_tiles + k_symv_reduce, half the bytes of the
    row-by-row kernel) once the level has 2048 unknowns: same result to FP64 rounding as k_gemv_f32x4 (FCVM_L2_SYMV=0) on a
    level whose size is not a multiple of the 128-wide tiles, the same PCG iteration count, and bitwise the same on a rerun."""
    from fcvm_hip import meshgen
    from fcvm_hip._lib import call
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    case = meshgen.cube_case(20, l=10.0, traction=(0.0, 0.0, -1.0))
    monkeypatch.setenv("FCVM_MG_TARGET", "12")     # 9^3 aggregates: 4374 unknowns = 34 tiles + 22 rows
    rng = np.random.default_rng(3)
    out = {}
    for name in ("rows", "symv", "symv again"):
        if name == "rows":
            monkeypatch.setenv("FCVM_L2_SYMV", "0")
        else:
            monkeypatch.delenv("FCVM_L2_SYMV", raising=False)
        m = GpuModel(case["elNodes"], case["nocoord"], (210000.0, 0.3, 7.8e-6), case["fix"],
                     Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctx=Context(0))
        c = m.ctx
        glv, modf, _, _ = m.calc_gsm((0.0, 0.0, -9.81), want_x=False)
        n2 = c.precond_info()["l2_unknowns"]
        assert n2 >= 2048 and n2 % 128 != 0
        if "r2" not in out:
            out["r2"] = rng.standard_normal(n2)
        r2, e2 = c.array(out["r2"]), c.empty(n2)
        call("fcvm_mg_l2_solve", c.h, r2.ptr, e2.ptr)
        f = c.mul(m.fixdof, glv)
        c.axpby(1.0, f, 1.0, modf, out=f)
        u, info = c.solve(f, rtol=1e-10)
        out[name] = (e2.get(), info["iterations"], u.get())
        c.close()
    assert rel(out["symv"][0], out["rows"][0]) < 1e-12
    assert abs(out["symv"][1] - out["rows"][1]) <= 1
    assert rel(out["symv"][2], out["rows"][2]) < 1e-8
    assert np.array_equal(out["symv"][0], out["symv again"][0]) and np.array_equal(out["symv"][2], out["symv again"][2])


@pytest.mark.parametrize("switch", ["FCVM_TOPO_EARLY=0", "FCVM_TRINV_ROWS=1", "FCVM_EMAP_HOST=1", "FCVM_L1_NT=1", "FCVM_MORTON_THREADS=1",
                                    "FCVM_WIN_SYM=0", "FCVM_WIN_SYM_LO=0", "FCVM_MORTON_AXIS_SCALE=1", "FCVM_ASSEMBLE_SYM=0", "FCVM_SYM_PAIR=0",
                                    "FCVM_ASM_LPR=8", "FCVM_ASM_LPR_LONG=16", "FCVM_ASM_LPR_LONG=4", "FCVM_WSYM_EAGER=1", "FCVM_TET4_LDS=0"])
def test_set_up_switches_do_not_change_the_solution(monkeypatch, switch):
    """The round-3 set-up paths against the ones they replaced: level-1 topology on the background thread / at the first
    assembly, triangular inverse by recursive doubling / block row by block row, element map from the device kernel / the host
    builder, plain / nontemporal loads of the level-1 matrix, Morton sort on all threads / one, the symmetric form of the window
    format / the full form (all three passes, the 16-bit passes only), Morton keys with one scale / per-axis scales; round 4: assembly
    through the sliced-ELL array and the copy kernel, lanes per row of the assembly into the symmetric form, its device half at
    fcvm_set_mesh, level-1 operator by the round-1 kernel.  Same matrix bitwise, same PCG iteration count (+-1 where the inverse is formed in another order), same solution."""
    from fcvm_hip import examples
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    case = examples.embankment(refine=2)
    out = []
    for env in (None, switch):
        if env:
            k, v = env.split("=")
            monkeypatch.setenv(k, v)
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(), ctx=Context(0))
        c = m.ctx
        glv, modf, _, _ = m.calc_gsm((0.0, 0.0, -10.0), want_x=False)
        f = c.mul(m.fixdof, glv)
        u, info = c.solve(f, rtol=1e-11)
        out.append((c.matrix_coo()[2], info["iterations"], u.get(), c.precond_info()["l2_aggregates"]))
        c.close()
    if "SYM" in switch and "EAGER" not in switch:
        # round 4: by default the matrix is assembled straight into the symmetric window form and read back from it -- K_ji is
        # the transpose of the stored K_ij to the bit -- while these switches go through the sliced-ELL array, where both halves
        # are summed separately (the same sums, rounded independently)
        assert np.abs(out[0][0] - out[1][0]).max() <= 1e-15 * np.abs(out[0][0]).max()
    elif switch != "FCVM_MORTON_AXIS_SCALE=1":       # (another row order: the same entries, assembled in the same order per entry)
        assert np.array_equal(out[0][0], out[1][0])
    assert out[0][3] == out[1][3] > 0
    assert abs(out[0][1] - out[1][1]) <= (2 if "SYM" in switch or "MORTON_AXIS" in switch else 1)
    assert rel(out[1][2], out[0][2]) < 1e-9


def test_a_second_mesh_on_one_context_while_the_background_builders_still_run(orc):
    """fcvm_set_mesh leaves two host threads behind (level-1 topology, symmetric window form); a second
    fcvm_set_mesh on the same context -- before anything waited for them -- must join them, drop their results and build the new
    mesh's; so must closing the context.  The second model then assembles (straight into the new symmetric form) and solves."""
    from fcvm_hip import examples
    from fcvm_hip.device import Context
    a, b = small_case(7), examples.embankment(refine=1)
    c = Context(0)
    c.set_mesh(np.asarray(a["elNodes"]), np.asarray(a["nocoord"], dtype=np.float64))
    c.set_mesh(np.asarray(b["elNodes"]), np.asarray(b["nocoord"], dtype=np.float64))      # (no finish_setup in between)
    c.set_material(*[float(v) for v in np.asarray(b["material"]).reshape(-1)[:3]])
    c.set_fix(b["fix"])
    n = 3 * len(b["nocoord"])
    glv, modf, V = c.assemble((0.0, 0.0, -10.0))
    info = c.spmv_sym_info()
    assert info["built"] and info["assembled_directly"]
    fixdof = np.ones(n); fixdof[list(b["fix"])] = 0.0
    f = c.array(fixdof * glv.get() + modf.get())
    u, it = c.solve(f, rtol=1e-10)
    r = c.axpby(1.0, f, -1.0, c.spmv(u, c.empty(n)))
    assert c.norm(r) <= 5e-10 * c.norm(f)
    c.close()
    c2 = Context(0)                     # closed right after set_mesh: the threads are joined by fcvm_destroy
    c2.set_mesh(np.asarray(a["elNodes"]), np.asarray(a["nocoord"], dtype=np.float64))
    c2.close()


def test_smoother_bounds_are_the_largest_eigenvalue_within_two_percent():
    """Round 4: every smoother interval / weight of the multigrid cycle comes from lambda_max(D^-1 A) of its level (D = the 3x3
    diagonal blocks).  Rounds 1-3 took 1.1 x fifteen power iterations, which reached 87-90 % of it -- the damped Jacobi step of cycle
    form 2 then sat at c0 * lambda_max = 1.90-2.02, 2 being where the cycle stops being positive definite.  The Lanczos estimate
    (mg_estimate_lmax) must be within 2 % below the eigenvalue (ARPACK on the matrix read back from the device), so that the 1.02
    margin holds what it promises."""
    import scipy.sparse.linalg as spla
    from fcvm_hip import examples
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    for case, grav in ((examples.embankment(refine=1), (0.0, 0.0, -10.0)), (small_case(7), (0.0, 0.0, -9.81))):
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                     Loads(ufaces=case["ufaces"], uloads=case["uloads"]) if "ufaces" in case else Loads())
        c = m.ctx
        m.calc_gsm(grav, want_x=False)
        K = gpu_matrix(m).tocsc()
        n = m.ndof
        D = sp.block_diag([K[3 * i:3 * i + 3, 3 * i:3 * i + 3].toarray() for i in range(n // 3)]).tocsc()
        lam = float(spla.eigsh(K, k=1, M=D, which="LM", return_eigenvectors=False, tol=1e-6)[0])
        est = c.precond_info()["lmax0"]
        assert 0.98 * lam <= est <= 1.0001 * lam, (est, lam)
        c.close()


def test_dense_spd_inverse_kernels(gpu_ctx):
    """fcvm_spd_inverse (blocked Cholesky, triangular inverse, X^T X in HIP kernels: the aggregate level's solver) against
    LAPACK, sizes around the 64-wide tiles."""
    from fcvm_hip._lib import call
    c = gpu_ctx
    rng = np.random.default_rng(11)
    for n in (5, 64, 130, 700, 1500):   # 1, 1, 3, 11, 24 tiles: one to five doubling levels, ragged last pairs
        q = rng.standard_normal((n, n))
        a = q @ q.T + n * np.diag(rng.uniform(0.5, 2.0, n))
        d_a, d_i = c.array(a.reshape(-1)), c.empty(n * n)
        call("fcvm_spd_inverse", c.h, n, d_a.ptr, d_i.ptr)
        inv = d_i.get().reshape(n, n)
        assert np.array_equal(inv, inv.T)
        assert np.abs(inv @ a - np.eye(n)).max() < 1e-11 * np.linalg.cond(a)
        assert rel(inv, np.linalg.inv(a)) < 1e-10
    # not positive definite -> error, not garbage
    bad = np.eye(70)
    bad[40, 40] = -1.0
    d_bad, d_out = c.array(bad.reshape(-1)), c.empty(70 * 70)
    with pytest.raises(Exception, match="positive definite"):
        call("fcvm_spd_inverse", c.h, 70, d_bad.ptr, d_out.ptr)


def test_vector_kernels(gpu_ctx):
    c = make_model(small_case(1)).ctx
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(1000), rng.standard_normal(1000)
    dx, dy = c.array(x), c.array(y)
    assert np.allclose(c.axpby(2.0, dx, -0.5, dy).get(), 2 * x - 0.5 * y, rtol=0, atol=1e-15)
    assert np.allclose(c.mul(dx, dy).get(), x * y, rtol=0, atol=1e-15)
    assert c.dot(dx, dy) == pytest.approx(float(x @ y), rel=1e-13)
    assert c.max_node_norm(dx, 300) == pytest.approx(np.sqrt((x[:900].reshape(-1, 3) ** 2).sum(1).max()), rel=1e-14)


def test_embankment_analysis_matches_reference_files(orc, embankment):
    """The whole GPU analysis (assembly, PCG, Riks loop, radial return, PEEQ/CSR, nodal mapping,
    principal stresses) against the reference's own Embankment_with_Ditch_Example results."""
    from fcvm_hip import inpout
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    from test_oracle_golden import sig3
    g, meta = embankment
    ctl = inpout.parse_inp_lines(meta["inp"])
    fix = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    res = run_analysis(g["elNodes"], g["nocoord"], g["material"], fix, Loads(), ctl, rtol=1e-11)
    tab = g["out_table"]
    assert len(res["lout"]) == len(tab)
    assert np.array_equal(res["crip"], tab[:, 0].astype(int))
    for col, val in ((4, res["lout"]), (5, res["un"]), (6, res["peeqplot"]), (7, res["pplot"]), (8, res["svmplot"]),
                     (9, res["triaxplot"]), (10, res["ecrplot"]), (11, res["csrplot"])):
        assert sig3(val, tab[:, col]), "column %d" % col
    tol = 1e-6  # north_star: 1e-6 relative on load-displacement curve and stresses
    assert rel(res["displacements"], g["vtk_Displacement"].reshape(-1)) < tol
    assert rel(res["tet10stress"], g["vtk_Stress_Tensor"]) < tol
    assert rel(res["tet10peeq"], g["vtk_Equivalent_Plastic_Strain"]) < tol
    assert rel(res["tet10csr"], g["vtk_Critical_Strain_Ratio"]) < tol
    assert rel(res["tet10svm"], g["vtk_von_Mises_Stress"]) < tol
    assert rel(res["principal"][:, 0], g["vtk_Major_Principal_Stress"]) < tol
    assert rel(res["principal"][:, 2], g["vtk_Minor_Principal_Stress"]) < tol
    # entry by entry too (every entry above 1e-3 of the field maximum), not only against the maximum
    each = {k: rel_each(a, b) for k, a, b in (("displacement", res["displacements"], g["vtk_Displacement"].reshape(-1)),
                                               ("stress", res["tet10stress"], g["vtk_Stress_Tensor"]),
                                               ("von Mises", res["tet10svm"], g["vtk_von_Mises_Stress"]),
                                               ("peeq", res["tet10peeq"], g["vtk_Equivalent_Plastic_Strain"]),
                                               ("major principal", res["principal"][:, 0], g["vtk_Major_Principal_Stress"]))}
    assert max(each.values()) < 1e-5, each
    # and against the oracle run of the same analysis: curve to 1e-6
    o = orc.run_analysis(g["elNodes"], g["nocoord"], g["material"], fix, orc.Loads(), orc.read_inp_lines(meta["inp"]))
    assert rel(res["lout"], o["lout"]) < tol and rel(res["un"], o["un"]) < tol


def test_lug_pressure_example_matches_reference_files():
    """The reference's Lug_Pressure_Example through the C-ABI: pressure on the two cylindrical faces of the hole
    (face integrals with the element normals, fcVM.py:649-672), disp_output = incremental, two passes of 30 steps
    (continuation, fcVM.py:1304-1308) -- every column of all 61 rows of the reference's .out and the f64 nodal fields
    of its .vtk (north_star tolerance 1e-6)."""
    from fcvm_hip import inpout
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    from test_oracle_golden import check_lug_against_reference_files, lug_pressure_case
    g, meta = load_golden("lug")
    fix, pf, pv, inp = lug_pressure_case(g, meta)
    ctl = inpout.parse_inp_lines(inp)
    res = run_analysis(g["elNodes"], g["nocoord"], g["material"], fix, Loads(pfaces=pf, pvals=pv), ctl, rtol=1e-11, max_passes=2)
    check_lug_against_reference_files(res, g, 1e-6, tol_each=1e-5)


def test_cube_analysis_matches_oracle(orc):
    """Traction-loaded cube with prescribed displacements (moving boundary => reaction-force
    output, fcVM.py:1169-1177, 1522-1523), hardening, 6 load steps."""
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    case = small_case(3)
    ctl = dict(sig_yield=30.0, grav_x=0.0, grav_y=0.0, grav_z=-9.81, nstep=6, iterat_max=20, error_max=1e-4,
               relax=1.2, scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25,
               Et_E=0.05, target_LF=10.0, csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLN", maxImp="0.0",
               ev1="1.0", ev2="0.0")
    o = orc.run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                         orc.Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctl)
    assert o["peeq"].max() > 0, "case must yield"
    res = run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                       Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctl, rtol=1e-12)
    assert len(res["lout"]) == len(o["lout"])
    assert rel(res["lout"], o["lout"]) < 1e-6 and rel(res["un"], o["un"]) < 1e-6
    assert rel(res["displacements"], o["displacements"]) < 1e-6
    assert rel(res["stresses"], o["stresses"]) < 1e-6
    assert rel(res["peeq"], o["peeq"]) < 1e-6
    # ... and entry by entry, for every entry above a thousandth of the field's maximum
    assert rel_each(res["displacements"], o["displacements"]) < 1e-5 and rel_each(res["stresses"], o["stresses"]) < 1e-5
    assert rel_each(res["lout"], o["lout"]) < 1e-6
    assert np.array_equal(res["crip"], o["crip"])


@pytest.mark.parametrize("cells", [26, 55], ids=["105k", "998k-bench-size"])
def test_size_independent_properties(gpu_ctx, cells):
    """Properties that need no oracle, at sizes the CPU restatement would take minutes / hours for (55 = the mesh of
    the benchmark, 998,250 tet10): K symmetric (<x,Ky> = <y,Kx>), rigid translations in the null space of the
    unconstrained K, elastic internal force == K du, PCG solution satisfies K u = f; the element map built on the device equals
    the host rule entry by entry."""
    from fcvm_hip import meshgen
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    el, xyz = meshgen.cube_tet10(cells, cells, cells, 50.0, 50.0, 50.0)
    mat = (210000.0, 0.3, 7.8e-6)
    m = GpuModel(el, xyz, mat, {}, Loads())
    c = m.ctx
    glv, modf, V, _ = m.calc_gsm((0.0, 0.0, -9.81))
    assert c.check_element_map() == (100 * m.ne, 0)       # the device-built element map, every entry, also at 1 M elements
    assert V == pytest.approx(50.0 ** 3, rel=1e-12)
    assert glv.get().reshape(-1, 3)[:, 2].sum() == pytest.approx(-9.81 * 7.8e-6 * 50.0 ** 3, rel=1e-11)
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(m.ndof), rng.standard_normal(m.ndof)
    dx, dy = c.array(x), c.array(y)
    kx, ky = c.spmv(dx, c.empty(m.ndof)), c.spmv(dy, c.empty(m.ndof))
    assert c.dot(dy, kx) == pytest.approx(c.dot(dx, ky), rel=1e-11)
    t = np.tile([1.0, -2.0, 0.5], m.nn)
    kt = c.spmv(c.array(t), c.empty(m.ndof)).get()
    assert np.abs(kt).max() < 1e-9 * np.abs(kx.get()).max()
    ne = m.ne
    du = rng.standard_normal(m.ndof) * 1e-6
    d_du = c.array(du)
    q = c.empty(m.ndof)
    c.update_stress_load(c.array(np.full(4 * ne, 1e30)), None, d_du, c.zeros(24 * ne), c.empty(24 * ne), c.empty(24 * ne),
                         q, 0.0, False, c.empty(4 * ne, np.uint8))
    kdu = c.spmv(d_du, c.empty(m.ndof)).get()
    assert rel(q.get(), kdu) < 1e-10
    # clamp the bottom, solve, check the residual of the returned solution with an independent SpMV
    fix = {3 * int(n) + k: 0.0 for n in meshgen.nodes_on_plane(xyz, 2, 0.0) for k in range(3)}
    m2 = GpuModel(el, xyz, mat, fix, Loads(), ctx=c)
    glv, modf, _, _ = m2.calc_gsm((0.0, 0.0, -9.81))
    f = c.mul(m2.fixdof, glv)
    u, info = c.solve(f, rtol=1e-10)
    r = c.axpby(1.0, f, -1.0, c.spmv(u, c.empty(m.ndof)))
    assert c.norm(r) <= 2e-10 * c.norm(f)
    assert info["iterations"] > 10


def test_refined_embankment_analysis_matches_oracle(orc, monkeypatch):
    """The reference's Embankment model refined once (5,272 tet10, unstructured, nu = 0.45), full
    30-step collapse analysis, three-level multigrid PCG on the GPU against the oracle's direct solver."""
    from fcvm_hip import examples
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    monkeypatch.setenv("FCVM_L2_MIN_COARSE", "500")  # exercise the aggregate level on this small mesh too
    case = examples.embankment(refine=1)
    assert len(case["elNodes"]) == 8 * 659
    o = orc.run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"], orc.Loads(),
                         orc.read_inp_lines([str(case["ctl"][k]) for k in __import__("fcvm_hip.inpout", fromlist=["x"]).INP_KEYS]))
    res = run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(), case["ctl"], rtol=1e-11)
    assert res["model"].ctx.precond_info()["l2_aggregates"] > 0
    assert len(res["lout"]) == len(o["lout"]) == 31
    assert rel(res["lout"], o["lout"]) < 1e-6 and rel(res["un"], o["un"]) < 1e-6
    assert np.array_equal(res["crip"], o["crip"])
    assert rel(res["displacements"], o["displacements"]) < 1e-6
    assert rel(res["stresses"], o["stresses"]) < 1e-6
    assert rel(res["peeq"], o["peeq"]) < 1e-6


def test_gnly_updated_lagrange_analysis_matches_oracle(orc):
    """gnl == "GNLY" without imperfection (no buckling pre-analysis): tangent re-assembly with the
    plastic D - pmat and updated coordinates every Newton iteration (calcTSM), stress convection,
    follower geometry -- GPU against the oracle's restatement of that branch."""
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    case = small_case(3, prescribed=False)
    ctl = dict(sig_yield=30.0, grav_x=0.0, grav_y=0.0, grav_z=-9.81, nstep=4, iterat_max=20, error_max=1e-5, relax=1.2,
               scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25, Et_E=0.05, target_LF=10.0,
               csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLY", maxImp="0.0", ev1="1.0", ev2="0.0")
    case["uloads"] = case["uloads"] * 4.0
    o = orc.run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                         orc.Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctl)
    assert o["peeq"].max() > 0
    res = run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                       Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctl, rtol=1e-12)
    assert len(res["lout"]) == len(o["lout"])
    assert rel(res["lout"], o["lout"]) < 1e-6 and rel(res["un"], o["un"]) < 1e-6
    assert rel(res["displacements"], o["displacements"]) < 1e-6
    assert rel(res["stresses"], o["stresses"]) < 1e-6
    assert rel(res["peeq"], o["peeq"]) < 1e-6


def test_gnly_follower_pressure_matches_oracle(orc):
    """gnl == "GNLY" with a PRESSURE load: calcTSM re-integrates the pressure on the deformed geometry at every tangent (fcVM.py:858-883; no shipped
    result file of the reference exercises it).  GPU against the oracle's restatement of that branch -- and the GPU path takes only the
    displacements of the loaded faces' nodes to the host (fcvm_pack_nodes / fcvm_scatter_add_nodes, round 5), not the whole vector: pressure on
    the top face of the cube plus a uniform shear traction on one side (a static load beside the follower one)."""
    from fcvm_hip import meshgen
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    case = small_case(3, prescribed=False)
    ctl = dict(sig_yield=30.0, grav_x=0.0, grav_y=0.0, grav_z=-9.81, nstep=4, iterat_max=20, error_max=1e-5, relax=1.2,
               scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25, Et_E=0.05, target_LF=10.0,
               csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLY", maxImp="0.0", ev1="1.0", ev2="0.0")
    top = case["ufaces"]
    side = meshgen.boundary_faces_on_plane(case["elNodes"], case["nocoord"], 0, 10.0)
    kw = dict(pfaces=top, pvals=np.full(len(top), -30.0), ufaces=side, uloads=np.tile([0.0, 6.0, 0.0], (len(side), 1)))
    o = orc.run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"], orc.Loads(**kw), ctl)
    assert o["peeq"].max() > 0 and np.abs(o["displacements"]).max() > 1e-3
    res = run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(**kw), ctl, rtol=1e-12)
    assert res["model"]._follow is not None and res["model"]._follow[1] is not False          # the sparse follower route ran, beside a static part
    assert len(res["lout"]) == len(o["lout"])
    assert rel(res["lout"], o["lout"]) < 1e-6 and rel(res["un"], o["un"]) < 1e-6
    assert rel(res["displacements"], o["displacements"]) < 1e-6
    assert rel(res["stresses"], o["stresses"]) < 1e-6
    assert rel(res["peeq"], o["peeq"]) < 1e-6
    assert rel_each(res["displacements"], o["displacements"]) < 1e-5


def test_gnly_start_directions_carried_over_a_tangent_refresh_save_iterations_not_accuracy(monkeypatch):
    """GNLY: the solutions of the previous tangent carried over as start directions (FCVM_GNLY_CARRY, products with the new matrix
    formed by fcvm_spmv2) against starting every first solve with a new tangent from zero: the same analysis (the PCG works on
    the true residual either way), fewer PCG iterations."""
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    case = small_case(6, prescribed=False)
    ctl = dict(sig_yield=30.0, grav_x=0.0, grav_y=0.0, grav_z=-9.81, nstep=4, iterat_max=20, error_max=1e-5, relax=1.2,
               scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25, Et_E=0.05, target_LF=10.0,
               csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLY", maxImp="0.0", ev1="1.0", ev2="0.0")
    case["uloads"] = case["uloads"] * 4.0
    out = {}
    for carry in ("0", "2"):
        monkeypatch.setenv("FCVM_GNLY_CARRY", carry)
        out[carry] = run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                                  Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctl, rtol=1e-12)
    a, b = out["2"], out["0"]
    assert b["peeq"].max() > 0 and len(a["lout"]) == len(b["lout"])
    assert rel(a["lout"], b["lout"]) < 1e-8 and rel(a["displacements"], b["displacements"]) < 1e-8 and rel(a["peeq"], b["peeq"]) < 1e-7
    ia, ib = a["stats"]["pcg_iterations"], b["stats"]["pcg_iterations"]
    print("PCG iterations of the analysis: carried directions %d, from zero %d" % (ia, ib))
    assert ia < ib


def test_embankment_document_in_reference_out_file_out(tmp_path):
    """Drop-in at file level: the reference's Embankment_with_Ditch_Example document (mesh, BREP, constraints) and its
    control file go in, the GPU path writes a .out file that is the reference's shipped one byte for byte and a .vtk
    whose f64 nodal fields match the reference's to 1e-6."""
    import os
    import fcVM as ft
    from conftest import REPO
    from fcvm_hip import inpout, vtkio
    g, meta = load_golden("embankment")
    with open(tmp_path / (meta["name"] + ".inp"), "w") as f:
        f.write("\n".join(meta["inp"]) + "\n")
    ft.run_fcstd(os.path.join(REPO, "tests", "golden", "embankment.FCStd"), str(tmp_path), str(tmp_path), name=meta["name"])
    mine = open(tmp_path / (meta["name"] + ".out")).read()
    assert mine == open(os.path.join(REPO, "tests", "golden", "embankment_reference.out")).read()
    pts, cells, data = vtkio.read_vtk(tmp_path / (meta["name"] + ".vtk"))
    assert np.array_equal(cells + 1, g["elNodes"])
    assert rel(data["Displacement"], g["vtk_Displacement"]) < 1e-6
    assert rel(data["Stress_Tensor"], g["vtk_Stress_Tensor"]) < 1e-6
    assert rel(data["Equivalent_Plastic_Strain\n"], g["vtk_Equivalent_Plastic_Strain"]) < 1e-6


@pytest.mark.parametrize("key", ["lug_faceload", "simple_shear", "block_disp", "elastica"])
def test_reference_documents_through_the_whole_chain(orc, tmp_path, key):
    """More of the reference's FreeCAD documents (with their shipped control files) through fcVM.run_fcstd on the GPU,
    against the oracle fed by the same reader: Lug_FaceLoad_Example (force on a cylindrical face, linear buckling run),
    Simple Shear (face forces, vertex supports), Block_Disp_Control_Example (prescribed displacements, reaction-force
    output, 17-line control file), Elastica (GNLY, 50 steps of large rotation with tangent re-assembly)."""
    import json
    import os
    import fcVM as ft
    from conftest import REPO
    from fcvm_hip import fcstd, inpout
    meta = json.load(open(os.path.join(REPO, "tests", "golden", "documents.json")))[key]
    with open(tmp_path / (key + ".inp"), "w") as f:
        f.write("\n".join(meta["inp"]) + "\n")
    src = os.path.join(REPO, "tests", "golden", key + ".FCStd")
    res = ft.run_fcstd(src, str(tmp_path), str(tmp_path), name=key)
    m = fcstd.load_model(src)
    L = m["loads"]
    ctl = inpout.parse_inp_lines(meta["inp"])
    o = orc.run_analysis(m["elNodes"], m["nocoord"], m["material"], m["fix"],
                         orc.Loads(pfaces=L.pfaces, pvals=L.pvals, ufaces=L.ufaces, uloads=L.uloads, edges=L.edges, eloads=L.eloads,
                                   vnodes=L.vnodes, vloads=L.vloads), dict(ctl), use_imperfection=True)
    assert len(res["lout"]) == len(o["lout"])
    if len(o["lout"]) > 1:
        assert rel(res["lout"], o["lout"]) < 1e-6
    assert rel(res["un"], o["un"]) < 1e-6
    assert rel(res["disp_el"], o["disp_el"]) < 1e-6 and rel(res["displacements"], o["displacements"]) < 1e-6
    if ctl["gnl"] == "GNLY" and ctl["nstep"] == 1:
        assert np.allclose(res["eigenval"], o["eigenval"], rtol=1e-6)
    else:
        assert rel(res["tet10stress"], o["tet10stress"]) < 1e-6
    assert os.path.getsize(tmp_path / (key + ".out")) > 0 and os.path.getsize(tmp_path / (key + ".vtk")) > 0


@pytest.mark.parametrize("sig_yield", [100.0, 20.0])
def test_vm_uniaxial_example_config0(orc, tmp_path, sig_yield):
    """BASELINE configs[0] end to end on the GPU: the reference's VM_Uniaxial_Tension_Example document and its
    (17-line) control file through fcVM.run_fcstd; against the oracle and the closed-form answers
    (elastic: stress diag(30,0,30) at load factor 3; sig_yield = 20: limit load factor 2)."""
    import json
    import os
    import fcVM as ft
    from conftest import REPO
    from fcvm_hip import fcstd, inpout
    meta = json.load(open(os.path.join(REPO, "tests", "golden", "vm_uniaxial.json")))
    lines = list(meta["inp"])
    lines[0] = str(sig_yield)
    with open(tmp_path / "vm.inp", "w") as f:
        f.write("\n".join(lines) + "\n")
    src = os.path.join(REPO, "tests", "golden", "vm_uniaxial.FCStd")
    res = ft.run_fcstd(src, str(tmp_path), str(tmp_path), name="vm")
    m = fcstd.load_model(src)
    L = m["loads"]
    o = orc.run_analysis(m["elNodes"], m["nocoord"], m["material"], m["fix"], orc.Loads(pfaces=L.pfaces, pvals=L.pvals),
                         inpout.parse_inp_lines(lines))
    assert len(res["lout"]) == len(o["lout"])
    assert rel(res["lout"], o["lout"]) < 1e-6 and rel(res["un"], o["un"]) < 1e-6
    assert rel(res["displacements"], o["displacements"]) < 1e-6
    assert rel(res["tet10stress"], o["tet10stress"]) < 1e-6 and rel(res["sigmises"], o["sigmises"]) < 1e-6
    if sig_yield == 100.0:
        assert res["lout"][-1] == 3.0
        assert np.abs(res["tet10stress"] - [30.0, 0.0, 30.0, 0.0, 0.0, 0.0]).max() < 1e-7
    else:
        assert abs(np.max(res["lout"]) - 2.0) < 3e-3 and rel(res["peeq"], o["peeq"]) < 1e-6
    assert os.path.getsize(tmp_path / "vm.out") > 0 and os.path.getsize(tmp_path / "vm.vtk") > 0


def test_fcVM_mirror_api_and_files(orc, tmp_path):
    """The reference-named entry points (fcVM.calcGSM / calcDisp / mapStresses / exportVTK with the
    reference's argument lists incl. the dummy first load rows) and the headless macro equivalent
    fcVM.run_fcstd: .FCStd + .inp in, .out + .vtk out; results against the oracle."""
    import os
    import fcVM as ft
    from conftest import REPO
    from fcvm_hip import fcstd, inpout, vtkio
    g, meta = load_golden("tensile")
    ctl = inpout.parse_inp_lines(meta["inp"])
    ctl["target_LF"] = 10.0
    inpout.write_inp(tmp_path / "tensile.inp", ctl)
    src = os.path.join(REPO, "tests", "golden", "tensile.FCStd")
    res = ft.run_fcstd(src, str(tmp_path), str(tmp_path), name="tensile")
    m = fcstd.load_model(src)
    L = m["loads"]
    o = orc.run_analysis(m["elNodes"], m["nocoord"], m["material"], m["fix"], orc.Loads(ufaces=L.ufaces, uloads=L.uloads),
                         orc.read_inp_lines([str(ctl[k]) for k in inpout.INP_KEYS]))
    assert rel(res["lout"], o["lout"]) < 1e-6 and rel(res["un"], o["un"]) < 1e-6
    rows = [ln.split() for ln in open(tmp_path / "tensile.out") if ln[:12].strip().isdigit()]
    assert len(rows) == len(o["lout"])
    assert np.allclose(np.array(rows, float)[:, 4], [float("%.2e" % v) for v in o["lout"]])
    pts, cells, data = vtkio.read_vtk(tmp_path / "tensile.vtk")
    assert np.array_equal(cells, m["elNodes"] - 1)
    assert rel(data["Displacement"].reshape(-1), o["displacements"]) < 1e-6
    assert rel(data["Stress_Tensor"], o["tet10stress"]) < 1e-6
    assert list(data)[:6] == ["Critical_Strain_Ratio\n", "Equivalent_Plastic_Strain\n", "von_Mises_Stress\n", "Triaxiality\n",
                              "Displacement", "Stress_Tensor"]
    # reference-style calls (numba-signature dummy rows in front of every load array)
    lf = np.vstack([[0] * 6, L.ufaces])
    fl = np.vstack([[0.0] * 3, L.uloads])
    dummy6, dummyp = np.array([[0] * 6]), np.array([0.0])
    args = (dummy6, dummyp, np.array([[0]]), np.array([[0.0, 0.0, 0.0]]), np.array([[0, 0, 0]]), np.array([[0.0, 0.0, 0.0]]), lf, fl)
    mat = np.array([m["material"]])
    stm, row, col, glv, modf, V, lsx, lsy, lsz, ne, nn, x = ft.calcGSM(m["elNodes"], m["nocoord"], mat, m["fix"], 0.0, 0.0, 0.0, *args)
    assert ne == 24 and nn == 63 and abs(V - 1000.0) < 1e-9 and abs(lsz - 1.0e5) < 1e-6
    sv, sr, sc = stm.to_coo()
    assert len(sv) > 0 and np.all(sr >= sc)
    fixdof = np.ones(3 * nn); fixdof[list(m["fix"])] = 0
    out = ft.calcDisp(m["elNodes"], m["nocoord"], fixdof, np.zeros(3 * nn), modf, mat, stm, row, col, glv, ctl["nstep"],
                      ctl["iterat_max"], ctl["error_max"], ctl["relax"], ctl["scale_re"], ctl["scale_up"], ctl["scale_dn"],
                      ctl["sig_yield"], ctl["disp_output"], ctl["ultimate_strain"], ft.HeadlessWindow(), ctl["Et_E"],
                      ctl["target_LF"], x, None, m["fix"], 0.0, 0.0, 0.0, *args, ctl["gnl"], ctl["maxImp"], ctl["ev1"], ctl["ev2"])
    assert len(out) == 19
    assert rel(out[0], o["displacements"]) < 1e-6 and rel(out[8], o["lout"]) < 1e-6 and rel(out[4], o["stresses"]) < 1e-6
    t10 = ft.mapStresses(False, m["elNodes"], m["nocoord"], out[4], out[5], out[6], out[7], None, ctl["sig_yield"])
    assert rel(t10[0], o["tet10stress"]) < 1e-6


def test_bar_buckling_gnly_matches_reference_files(orc):
    """GNLY on the GPU against the reference's own 'Bar buckling' results (imperfect geometry as
    exported by the reference): .out table and f64 nodal fields."""
    from fcvm_hip import inpout
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    from test_oracle_golden import sig3
    g, meta = load_golden("barbuckling")
    fix = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    ctl = inpout.parse_inp_lines(meta["inp"])
    ctl["buckling"] = False   # the exported coordinates already carry the reference's imperfection
    res = run_analysis(g["elNodes"], g["nocoord"], g["material"], fix, Loads(ufaces=g["ufaces"], uloads=g["uloads"]), ctl,
                       rtol=1e-12)
    tab = g["out_table"]
    assert len(res["lout"]) == len(tab)
    for col, val in ((4, res["lout"]), (5, res["un"]), (7, res["pplot"]), (8, res["svmplot"]), (9, res["triaxplot"])):
        assert sig3(val, tab[:, col]), "column %d" % col
    assert rel(res["displacements"], g["vtk_Displacement"].reshape(-1)) < 1e-6
    assert rel(res["disp_el"], g["vtk_Elastic_Displacement"].reshape(-1)) < 1e-6
    assert rel(res["tet10stress"], g["vtk_Stress_Tensor"]) < 1e-6
    each = {k: rel_each(a, b) for k, a, b in (("displacement", res["displacements"], g["vtk_Displacement"].reshape(-1)),
                                               ("elastic displacement", res["disp_el"], g["vtk_Elastic_Displacement"].reshape(-1)),
                                               ("stress", res["tet10stress"], g["vtk_Stress_Tensor"]))}
    assert max(each.values()) < 1e-5, each        # entry by entry (entries above 1e-3 of the field maximum)


def test_buckling_factors_on_gpu(orc):
    """GPU buckling pre-analysis (subspace iteration on the shifted operator, multigrid-PCG solves)
    against (a) the reference's printed factors of its 'Bar buckling' model and (b) the oracle's ARPACK
    restatement on a column with a rectangular (non-degenerate) cross-section."""
    from fcvm_hip import buckling, meshgen
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    g, meta = load_golden("barbuckling")
    fix = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    m = GpuModel(g["elNodes"], g["nocoord0"], g["material"], fix, Loads(ufaces=g["ufaces"], uloads=g["uloads"]))
    lam, vec, ue = buckling.buckling_factors(m, (0.0, 0.0, 0.0))
    assert np.allclose(np.sort(lam), g["buckling_factors"], rtol=0, atol=6e-6)
    # (b) 40 x 6 x 10 column, clamped at x = 0, compressed at x = L: two distinct bending modes
    el, xyz = meshgen.cube_tet10(10, 2, 3, 40.0, 6.0, 10.0)
    fixc = {3 * int(n) + k: 0.0 for n in meshgen.nodes_on_plane(xyz, 0, 0.0) for k in range(3)}
    faces = meshgen.boundary_faces_on_plane(el, xyz, 0, 40.0)
    ul = np.tile([-5000.0, 0.0, 0.0], (len(faces), 1))
    mat = np.array([210000.0, 0.3, 7.8e-6])
    w, v = orc.buckling_analysis(el, xyz, mat, fixc, (0.0, 0.0, 0.0), orc.Loads(ufaces=faces, uloads=ul))
    m2 = GpuModel(el, xyz, mat, fixc, Loads(ufaces=faces, uloads=ul))
    lam2, vec2, _ = buckling.buckling_factors(m2, (0.0, 0.0, 0.0))
    assert np.allclose(np.sort(lam2), np.sort(w), rtol=1e-7)
    for j in range(2):  # same shapes up to scale / sign
        a, b = vec2[:, np.argsort(lam2)[j]], v[:, np.argsort(w)[j]]
        cos = abs(a @ b) / np.linalg.norm(a) / np.linalg.norm(b)
        assert cos > 1 - 1e-6


def test_geometric_stiffness_rows_in_lds_equal_the_read_modify_write_kernel(monkeypatch):
    """K_g of the buckling pre-analysis (fcVM.py:1002-1006): the kernel that sums each row's blocks in LDS and writes them once
    against the round-1 kernel that adds every element's share to global memory (FCVM_GEOM_LDS=0), through K_g x on a mesh with
    odd and even slice lengths and a stress field that differs in every Gauss point."""
    import time
    from fcvm_hip import meshgen
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    el, xyz = meshgen.cube_tet10(9, 7, 5, 9.0, 7.0, 5.0)
    fixed = {3 * int(n) + k: 0.0 for n in meshgen.nodes_on_plane(xyz, 0, 0.0) for k in range(3)}
    xyz = xyz + 0.03 * np.random.default_rng(5).standard_normal(xyz.shape)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FCVM_GEOM_LDS", flag)
        m = GpuModel(el, xyz, np.array([210000.0, 0.3, 7.8e-6]), fixed, Loads(), ctx=Context(0))
        c = m.ctx
        m.calc_gsm((0.0, 0.0, -10.0), want_x=False)
        sig = c.array(np.random.default_rng(6).standard_normal(24 * len(el)))
        c.assemble_buckling(sig, 0.1)
        c.sync()
        t0 = time.perf_counter()
        c.assemble_buckling(sig, 0.1)
        c.sync()
        dt = time.perf_counter() - t0
        x = c.array(np.random.default_rng(7).standard_normal(3 * len(xyz)))
        out[flag] = (c.spmv_geometric(x, c.empty(3 * len(xyz))).get(), dt)
        c.close()
    a, b = out["1"][0], out["0"][0]
    print("K_b and K_g assembly: rows in lds %.3f ms, global adds %.3f ms" % (1e3 * out["1"][1], 1e3 * out["0"][1]))
    assert np.linalg.norm(b) > 0
    assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max()


def test_gnly_with_buckling_imperfection_full_flow(orc, tmp_path):
    """The complete GNLY flow on the GPU -- elastic solution, buckling factors, imperfection from the
    buckling shapes, restart on the imperfect geometry, non-linear run, .out / .vtk with the buckling
    fields -- (a) on a column with distinct buckling modes against the oracle's ARPACK-based
    restatement, (b) on the reference's Bar buckling model from its PERFECT geometry: the direction
    of the imperfection inside the double eigenvalue is arbitrary (ARPACK start vector in the
    reference), so its load-displacement curve is met to ~1 % only."""
    import fcVM as ft
    from fcvm_hip import inpout, meshgen, vtkio
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    el, xyz = meshgen.cube_tet10(10, 2, 3, 40.0, 6.0, 10.0)
    fix = {3 * int(n) + k: 0.0 for n in meshgen.nodes_on_plane(xyz, 0, 0.0) for k in range(3)}
    faces = meshgen.boundary_faces_on_plane(el, xyz, 0, 40.0)
    ul = np.tile([-5000.0, 0.0, 0.0], (len(faces), 1))
    mat = np.array([210000.0, 0.3, 7.8e-6])
    ctl = dict(sig_yield=240.0, grav_x=0.0, grav_y=0.0, grav_z=0.0, nstep=5, iterat_max=20, error_max=1e-4, relax=1.2,
               scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25, Et_E=0.0, target_LF=10.0,
               csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLY", maxImp="0.5", ev1="1.0", ev2="0.3")
    o = orc.run_analysis(el, xyz, mat, fix, orc.Loads(ufaces=faces, uloads=ul), ctl, use_imperfection=True)
    inpout.write_inp(tmp_path / "column.inp", ctl)
    res = ft.run("column", el, xyz.copy(), [mat], fix, Loads(ufaces=faces, uloads=ul), str(tmp_path), str(tmp_path))
    assert np.allclose(np.sort(res["eigenval"]), np.sort(o["eigenval"]), rtol=1e-7)
    assert rel(res["nocoord"], o["nocoord"]) < 1e-7               # same imperfection
    assert len(res["lout"]) == len(o["lout"])
    assert rel(res["lout"], o["lout"]) < 1e-5 and rel(res["un"], o["un"]) < 1e-5
    assert rel(res["displacements"], o["displacements"]) < 1e-5 and rel(res["stresses"], o["stresses"]) < 1e-5
    head = open(tmp_path / "column.out").read()
    assert "analysis type:" in head and "elastic-plastic, geometric non-linear" in head and "elastic buckling factors:" in head
    pts, cells, data = vtkio.read_vtk(tmp_path / "column.vtk")
    names = list(data)
    assert names[4:8] == ["Displacement", "Elastic_Displacement", "Buckling_shape_for_lambda1_=_%s" % round(float(res["eigenval"][0]), 3),
                          "Buckling_shape_for_lambda2_=_%s" % round(float(res["eigenval"][1]), 3)]
    assert rel(pts, o["nocoord"]) < 1e-7
    # (b) the reference's bar from its perfect geometry
    g, meta = load_golden("barbuckling")
    fixb = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    r = run_analysis(g["elNodes"], g["nocoord0"], g["material"], fixb, Loads(ufaces=g["ufaces"], uloads=g["uloads"]),
                     inpout.parse_inp_lines(meta["inp"]))
    assert np.allclose(np.sort(r["eigenval"]), g["buckling_factors"], atol=6e-6)
    assert np.abs(r["nocoord"] - g["nocoord0"]).max() == pytest.approx(10.0, rel=1e-9)
    tab = g["out_table"]
    assert len(r["lout"]) == len(tab)
    assert np.allclose(r["lout"], tab[:, 4], rtol=2e-2, atol=1e-3) and np.allclose(r["un"], tab[:, 5], rtol=2e-2, atol=1e-2)


def test_u100_column_full_gnly_flow_matches_reference_files():
    """The reference's buckling_U100_3D run from its perfect geometry, entirely on the GPU: buckling factors of the .out
    header, buckling shapes / imperfect coordinates of the .vtk, and every row of the load / displacement table over the
    three passes of the reference's session (gravity + two end forces, GNLY with tangent re-assembly, post-peak branch)."""
    from fcvm_hip import inpout
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    from test_oracle_golden import sig3
    g, meta = load_golden("u100")
    fix = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    res = run_analysis(g["elNodes"], g["nocoord0"], g["material"], fix, Loads(ufaces=g["ufaces"], uloads=g["uloads"]),
                       inpout.parse_inp_lines(meta["inp"]), rtol=1e-12, max_passes=3, pass_targets=[1.0, 1000.0])
    assert np.allclose(res["eigenval"], g["buckling_factors"], rtol=0, atol=6e-6)
    for i, key in enumerate(("shape1", "shape2")):
        r, x = g[key].reshape(-1), res["eigenvec"][:, i]
        assert np.abs((x @ r) / (x @ x) * x - r).max() < 1e-6
    assert np.abs(res["nocoord"] - g["nocoord"]).max() < 1e-5
    tab = g["out_table_full"]   # three passes: to 0.8, to 1.0 (new target typed into the plot window), over the peak load
    assert len(res["lout"]) == len(tab) == 22 and res["lout"][8] == 0.8 and res["lout"][11] == 1.0
    assert np.array_equal(res["crip"], tab[:, 0].astype(int))
    for col, val in ((4, res["lout"]), (5, res["un"]), (6, res["peeqplot"]), (7, res["pplot"]), (8, res["svmplot"]),
                     (9, res["triaxplot"]), (10, res["ecrplot"]), (11, res["csrplot"])):
        assert sig3(val, tab[:, col]), "column %d" % col
    # calcSum (fcVM.py:2830-2900) over the selection of the shipped .avr (Edge7 / Face2): the GPU's final nodal fields average
    # to what the nodal fields of the reference's .vtk do (see test_calcSum_on_the_reference_avr_selection for the file itself)
    import fcVM as ft
    sel = ([g["avr_edge7"].tolist()], [g["avr_face2"].tolist()], g["nocoord0"])
    want = ft.calcSum(*sel, g["vtk_csr"], g["vtk_peeq"], g["vtk_svm"])
    got = ft.calcSum(*sel, res["tet10csr"], res["tet10peeq"], res["tet10svm"])
    for w, v in zip(want, got):
        assert abs(v[0] - w[0]) <= 1e-6 * max(abs(w[0]), 1e-3), (w, v)


def test_plate_with_hole_configs1_and_north_star_sizes():
    """BASELINE configs[1] (Plate_with_hole_Example, ~50k tet10, reference control file, 10 arc-length steps to collapse)
    and the north-star size (1.0 M tet10) on the GPU.  No oracle at these sizes; checked are the physics the
    reference's own .out shows (load history of its 395-element mesh within 5 % row by row, same plateau) and mesh
    independence of the collapse load (50k vs 1M within 0.3 %)."""
    import json
    import os
    from conftest import REPO
    from fcvm_hip import examples
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel, calc_disp
    ref = json.load(open(os.path.join(REPO, "tests", "golden", "plate_with_hole.json")))
    tab = np.array(ref["out_table"])
    out = {}
    for h in (4.0, 1.45):
        case = examples.plate_with_hole(h)
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
        res = calc_disp(m, case["ctl"])
        out[h] = (m.ne, np.asarray(res["lout"]), np.asarray(res["un"]))
        assert np.allclose(res["loadsum"], ref["loadsum"], atol=1e-6)
        m.ctx.close()
    (ne_a, la, ua), (ne_b, lb, ub) = out[4.0], out[1.45]
    assert 45000 < ne_a < 55000 and 0.95e6 < ne_b < 1.1e6
    assert len(la) == len(lb) == len(tab) == 11
    assert np.abs(la[1:] / tab[1:, 4] - 1.0).max() < 0.05 and np.abs(ua[1:4] / tab[1:4, 5] - 1.0).max() < 0.02
    assert abs(la[-1] / lb[-1] - 1.0) < 3e-3 and 0.215 < lb[-1] < 0.23


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the driver's keys plus the roofline / cpu_baseline objects (tiny workload)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--plate-h", "8", "--cpu-plate-h", "10", "--cells", "8", "--steps", "3",
                        "--warmup", "2"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["unit"] == "elements*GP/s" and "workload" in d["config"] and "model" not in d["config"]
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb and "CHOLMOD" in cb["solver_note"]
    # the timed region is cut from the real collapse analysis: same control file, same load history as the whole run
    ns = d["north_star_workload"]
    assert ns["steps"] == 10 and 0.2 < ns["collapse_load_factor"] < 0.26 and ns["newton_iterations"] >= 5
    assert abs(d["value"] - d["newton_iterations_per_s"] * 4 * float(d["config"]["workload"].split(": ")[1].split(" tet10")[0])) < 1e-6 * d["value"]


def test_command_line_entry_point(tmp_path):
    """python -m fcvm_hip <document> --control DIR --output DIR: the reference's Embankment document and control file in,
    the reference's .out file out (model name = the document's label, as in the macro)."""
    import os
    import shutil
    import subprocess
    import sys
    from conftest import REPO
    g, meta = load_golden("embankment")
    with open(tmp_path / (meta["name"] + ".inp"), "w") as f:
        f.write("\n".join(meta["inp"]) + "\n")
    doc = tmp_path / "model.FCStd"
    shutil.copyfile(os.path.join(REPO, "tests", "golden", "embankment.FCStd"), doc)
    env = dict(os.environ, PYTHONPATH=os.path.join(REPO, "fcvm-workbench_amd"))
    r = subprocess.run([sys.executable, "-m", "fcvm_hip", str(doc), "--control", str(tmp_path), "--output", str(tmp_path)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(tmp_path / (meta["name"] + ".out")).read() == open(os.path.join(REPO, "tests", "golden", "embankment_reference.out")).read()


def test_reruns_are_bitwise_identical():
    """No floating-point atomics anywhere on the path (row-owner assembly, per-element internal forces gathered in a fixed
    order, fixed-order reductions in the solver): two runs of an elastoplastic analysis in fresh contexts give bitwise
    identical load histories, displacements, stresses and iteration counts."""
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import run_analysis
    case = small_case(5, prescribed=False)
    ctl = dict(sig_yield=4.0, grav_x=0.0, grav_y=0.0, grav_z=-9.81, nstep=4, iterat_max=20, error_max=1e-4, relax=1.2,
               scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25, Et_E=0.02, target_LF=10.0,
               csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLN", maxImp="0.0", ev1="1.0", ev2="0.0")
    runs = []
    for _ in range(2):
        r = run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                         Loads(ufaces=case["ufaces"], uloads=case["uloads"]), dict(ctl))
        runs.append(r)
        r["model"].ctx.close()
    a, b = runs
    assert a["peeq"].max() > 0
    for k in ("lout", "un", "displacements", "stresses", "peeq", "tet10stress"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    assert a["stats"]["pcg_iterations"] == b["stats"]["pcg_iterations"]


def test_group_map_and_placement_tuning_do_not_change_a_number():
    """Round 5: (a) the order in which the workgroups of the symmetric-form kernels take the groups of the window format (fcvm_set_group_map:
    an eighth of the groups per XCD / identity / chunks of C groups per XCD, default 16) is a scheduling choice -- product, inner-product
    partials and 16-bit pass are bitwise the same under every map, also where the group count is no multiple of 8 C; (b) the placement
    tuning of the first assembly (fcvm_placement_info: the PCG's result vectors timed against fresh allocations on the assembled matrix)
    moves buffers, not numbers: a context with FCVM_TUNE_PLACEMENT=1 (off) solves to the same bits."""
    import ctypes as C
    from fcvm_hip._lib import call
    case = small_case(9, prescribed=False)          # 4374 tet10 / 6859 nodes: 14 groups, not a multiple of 8
    m = make_model(case)
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm((0, 0, -9.81))
    si = c.spmv_sym_info()
    assert si["built"] and si["in_use"], si
    x = c.array(np.random.default_rng(4).standard_normal(m.ndof))
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    ref = None
    for mode in (16, 0, 1, 2, 3, 64):
        call("fcvm_set_group_map", c.h, mode)
        y = c.spmv(x, c.empty(m.ndof)).get()
        z = c.spmv_lo(x, c.empty(m.ndof)).get()
        out = c.empty(8)
        yd = c.empty(m.ndof)
        call("fcvm_spmv_dot", c.h, x.ptr, yd.ptr, out.ptr)
        u, info = c.solve(f, rtol=1e-11)
        got = (y, z, yd.get(), out.get()[:1], u.get(), info["iterations"])
        if ref is None:
            ref = got
        else:
            for a, b in zip(ref[:5], got[:5]):
                assert np.array_equal(a, b), mode
            assert got[5] == ref[5]
    call("fcvm_set_group_map", c.h, 16)
    pi = c.placement_info()
    assert not pi["tuned"]                          # (a matrix this small lives in the caches: nothing to tune)
    os.environ["FCVM_TUNE_PLACEMENT"] = "1"
    try:
        m2 = make_model(case)
    finally:
        del os.environ["FCVM_TUNE_PLACEMENT"]
    glv2, modf2, _, _ = m2.calc_gsm((0, 0, -9.81))
    f2 = m2.ctx.mul(m2.fixdof, glv2)
    m2.ctx.axpby(1.0, f2, 1.0, modf2, out=f2)
    u2, info2 = m2.ctx.solve(f2, rtol=1e-11)
    assert np.array_equal(u2.get(), ref[4]) and info2["iterations"] == ref[5]


def test_boundary_groups_first_is_bitwise_the_unsplit_pass():
    """Multi-GPU layer, round 5: the three fine-level passes launch the groups of the window format that hold rows of shared nodes first
    (fcvm_set_interface_nodes, fcvm_spmv_dot_part / fcvm_spmv_lo_part part 1), so that the shared-node exchange can start while the interior
    groups (part 2) still run.  Here on one GPU: for a made-up interface (the nodes of a plane through the cube, and a scattered set) part 1
    leaves every interface entry complete, parts 1 + 2 are bitwise the unsplit product -- vector, inner product, 16-bit pass."""
    import ctypes as C
    from fcvm_hip import meshgen
    from fcvm_hip._lib import call
    case = small_case(12, prescribed=False)         # 10,368 tet10 / 15,625 nodes: ~31 groups
    m = make_model(case)
    c = m.ctx
    m.calc_gsm((0, 0, -9.81))
    assert c.spmv_sym_info()["in_use"]
    x = c.array(np.random.default_rng(7).standard_normal(m.ndof))
    out_ref, out = c.empty(8), c.empty(8)
    y_ref, z_ref = c.empty(m.ndof), c.empty(m.ndof)
    call("fcvm_spmv_dot_part", c.h, x.ptr, y_ref.ptr, out_ref.ptr, 0)
    call("fcvm_spmv_lo_part", c.h, x.ptr, z_ref.ptr, 0)
    assert np.array_equal(y_ref.get(), c.spmv(x, c.empty(m.ndof)).get()) and np.array_equal(z_ref.get(), c.spmv_lo(x, c.empty(m.ndof)).get())
    plane = meshgen.nodes_on_plane(case["nocoord"], 0, 5.0)
    scattered = np.random.default_rng(8).choice(m.nn, 40, replace=False)
    for nodes in (plane, scattered):
        nodes = np.ascontiguousarray(nodes, np.int64)
        o = (C.c_int64 * 2)()
        call("fcvm_set_interface_nodes", c.h, len(nodes), nodes.ctypes.data, o)
        assert o[0] > 0 and o[1] > 0 and o[0] + o[1] == c.spmv_info()["groups"], tuple(o)
        dofs = (3 * nodes[:, None] + np.arange(3)[None, :]).reshape(-1)
        for name, ref in (("dot", y_ref), ("lo", z_ref)):
            y = c.array(np.full(m.ndof, np.nan))
            if name == "dot":
                call("fcvm_spmv_dot_part", c.h, x.ptr, y.ptr, out.ptr, 1)
            else:
                call("fcvm_spmv_lo_part", c.h, x.ptr, y.ptr, 1)
            yh = y.get()
            assert np.array_equal(yh[dofs], ref.get()[dofs]), name                  # every shared entry is there after part 1
            assert np.isnan(yh).any()                                               # ... and the interior is not yet
            if name == "dot":
                call("fcvm_spmv_dot_part", c.h, x.ptr, y.ptr, out.ptr, 2)
                assert np.array_equal(out.get()[:1], out_ref.get()[:1])
            else:
                call("fcvm_spmv_lo_part", c.h, x.ptr, y.ptr, 2)
            assert np.array_equal(y.get(), ref.get()), name
    call("fcvm_set_interface_nodes", c.h, 0, None, None)


def test_fused_inner_products_are_bitwise_the_single_ones():
    """fcvm_vec_mdot (the Newton driver's projection / Gram / Riks scalars in one pass) returns, for every pair, exactly the value
    of fcvm_vec_dot: same grid, same per-thread order, same two-stage sums -- sizes around the block and grid-stride limits,
    1 to 17 vectors (more than 16 go in batches)."""
    c = make_model(small_case(2)).ctx       # (the reductions use the work buffers of a context with a mesh)
    rng = np.random.default_rng(11)
    for n in (1, 255, 256, 257, 70001, 256 * 2048 + 3):
        y = c.array(rng.standard_normal(n))
        for k in (1, 3, 16, 17):
            xs = [c.array(rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4)) for _ in range(k)]
            fused = c.mdot(xs, y)
            single = [c.dot(x, y) for x in xs]
            assert fused == single, (n, k)
            assert np.allclose(fused, [float(x.get() @ y.get()) for x in xs], rtol=1e-12, atol=1e-12 * n)


def test_closing_a_context_returns_its_device_memory(gpu_ctx):
    """Arrays that are still referenced when their context closes (load vectors, solutions kept by the caller) used to keep
    their device memory for good: 30 MB per 83k-element model, ~10 MB per model of this test.  Device memory in use after 6 and
    after 14 models must agree (the runtime's own pool settles during the first few)."""
    keep = []

    def one():
        m = make_model(small_case(16))
        glv, modf, _, _ = m.calc_gsm((0.0, 0.0, -9.81), want_x=False)
        u, _ = m.ctx.solve(m.ctx.mul(m.fixdof, glv), rtol=1e-8)
        keep.append((glv, modf, u))          # still alive after close()
        m.ctx.close()

    def used():
        free, total = gpu_ctx.mem_info()
        return total - free

    for _ in range(6):
        one()
    a = used()
    for _ in range(8):
        one()
    assert used() - a < 24 * 2 ** 20, (a, used())


def test_error_paths_and_mirror_kernels(orc, gpu_ctx):
    """Loud failures (no silent fallback) and the array-level mirrors fcVM.update_stress_load /
    update_PEEQ_CSR / calculate_principal_stress."""
    import fcVM as ft
    from fcvm_hip._lib import FcvmError
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    c = Context(0)
    with pytest.raises(FcvmError, match="assemble"):
        c.solve(c.zeros(3), rtol=1e-8)                      # no mesh / matrix yet
    case = small_case(2)
    el_bad = case["elNodes"].copy()
    el_bad[0, 0] = len(case["nocoord"]) + 5
    with pytest.raises(FcvmError, match="out of range"):
        c.set_mesh(el_bad, case["nocoord"])
    c.set_mesh(case["elNodes"], case["nocoord"])
    with pytest.raises(FcvmError, match="nu"):
        c.set_material(210000.0, 0.5, 1.0)
    with pytest.raises(FcvmError, match="dof out of range"):
        c.set_fix({3 * len(case["nocoord"]) + 1: 0.0})
    c.set_material(210000.0, 0.3, 7.8e-6)
    f = c.array(np.ones(3 * len(case["nocoord"])))
    with pytest.raises(FcvmError, match="assemble"):
        c.solve(f)
    # unconstrained body: K is singular, the rhs is not self-equilibrated -> PCG must report, not return garbage
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], {}, Loads())
    m.ctx.set_precond("bj")
    glv, modf, _, _ = m.calc_gsm((0.0, 0.0, -9.81))
    with pytest.raises(FcvmError, match="PCG"):
        m.ctx.solve(glv, rtol=1e-12, maxit=200)
    # array-level mirrors
    rng = np.random.default_rng(4)
    ne, nn = len(case["elNodes"]), len(case["nocoord"])
    du, disp = rng.standard_normal(3 * nn) * 2e-3, np.zeros(3 * nn)
    sig, sy = rng.standard_normal(24 * ne) * 40, np.full(4 * ne, 220.0)
    mat = np.array([case["material"]])
    out = [np.zeros(24 * ne), np.zeros(24 * ne), np.ones(3 * nn), np.zeros(4 * ne, bool)]
    ft.update_stress_load(ft.gaussPoints()[0], case["elNodes"], case["nocoord"], mat, sy, disp, du, sig, out[0], out[1], out[2],
                          0.0, False, out[3])
    ref = [np.zeros(24 * ne), np.zeros(24 * ne), np.ones(3 * nn), np.zeros(4 * ne, np.uint8)]
    orc.update_stress_load(case["elNodes"], case["nocoord"], case["material"], sy, disp, du, sig, ref[0], ref[1], ref[2], 0.0,
                           False, ref[3])
    assert rel(out[0], ref[0]) < KTOL and rel(out[1], ref[1]) < KTOL and rel(out[2], ref[2]) < KTOL   # qin is ADDED to
    assert np.array_equal(out[3].astype(np.uint8), ref[3])
    peeq, csr, tri, pre, svm, ecr = (np.zeros(4 * ne) for _ in range(6))
    sy2 = sy.copy()
    ft.update_PEEQ_CSR(ne, mat, out[1], out[0], sy2, 0.25, peeq, csr, tri, pre, svm, ecr, 0.0)
    p2, c2, t2, pr2, s2, e2 = (np.zeros(4 * ne) for _ in range(6))
    sy3 = sy.copy()
    orc.update_PEEQ_CSR(ne, case["material"], ref[1], ref[0], sy3, 0.25, p2, c2, t2, pr2, s2, e2, 0.0)
    assert rel(peeq, p2) < 1e-12 and rel(csr, c2) < 1e-12 and rel(svm, s2) < 1e-12 and rel(ecr, e2) < 1e-12
    t10 = rng.standard_normal((nn, 6))
    got = ft.calculate_principal_stress(t10)
    want = orc.calculate_principal_stress(t10)
    for a, b in zip(got[:3], want[:3]):
        assert rel(a, b) < 1e-12


def _vertex_patches(el, nn):
    """node lists of the vertex patches of a tet10 mesh: every corner node with the mid-side nodes of its edges (numpy restatement of
    the tables the library builds in mg_topology_host)"""
    from fcvm_hip import meshgen
    e0 = np.asarray(el) - 1
    star = {}
    for k, (a, b) in enumerate(meshgen.MID_PAIRS):
        for m, ca, cb in zip(e0[:, 4 + k], e0[:, a], e0[:, b]):
            star.setdefault(int(ca), set()).add(int(m))
            star.setdefault(int(cb), set()).add(int(m))
    for c in np.unique(e0[:, :4]):
        star.setdefault(int(c), set())
    return [[c] + sorted(ms) for c, ms in sorted(star.items())]


@pytest.mark.parametrize("name", ["plate", "embankment"])
def test_vertex_patch_smoother_is_the_additive_schwarz_operator_and_cuts_the_iterations(monkeypatch, name):
    """Round 5 (FCVM_MG_PATCH=1; off by default): the fine level of the cycle smoothed by the additive Schwarz operator over
    the vertex patches, S = sum_c R_c^T (R_c K R_c^T)^-1 R_c, in the place of the damped block-Jacobi step.  (i) fcvm_debug_patch_apply against that sum formed with numpy from
    the matrix the library holds (FP32 inverses: 1e-5); (ii) the same solution as with block-Jacobi in at most 0.9 x the PCG
    iterations (small meshes: the coarse solve is nearly exact there); (iii) the preconditioner is symmetric: x.M^-1 y = y.M^-1 x."""
    from fcvm_hip import examples
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    case = examples.plate_with_hole(8.0) if name == "plate" else examples.embankment(refine=1)
    ld = Loads(ufaces=case["ufaces"], uloads=case["uloads"]) if name == "plate" else Loads()
    grav = (0.0, 0.0, 0.0) if name == "plate" else (0.0, 0.0, -10.0)
    res = {}
    monkeypatch.setenv("FCVM_VERBOSE", "1")     # (what the library says about the patches goes into the report of a failure)
    for mode in ("patch", "jacobi"):
        monkeypatch.setenv("FCVM_MG_PATCH", "1" if mode == "patch" else "0")
        print("leg:", mode, flush=True)
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], ld, ctx=Context(0))
        c = m.ctx
        glv, modf, _, _ = m.calc_gsm(grav, want_x=False)
        f = c.axpby(1.0, c.mul(m.fixdof, glv), 1.0, modf)
        info_p = c.patch_info()
        assert info_p["in_use"] == (mode == "patch") and c.precond_info()["patch_smoother"] == (mode == "patch")
        if mode == "patch":
            assert info_p["patches"] == len(np.unique(np.asarray(case["elNodes"])[:, :4])) and info_p["lmax"] > 1.0
            K = gpu_matrix(m)
            rng = np.random.default_rng(5)
            x = rng.standard_normal(m.ndof)
            want = np.zeros(m.ndof)
            for nodes in _vertex_patches(case["elNodes"], m.nn):
                idx = (3 * np.asarray(nodes)[:, None] + np.arange(3)[None, :]).ravel()
                want[idx] += np.linalg.solve(K[idx][:, idx].toarray(), x[idx])
            got = c.patch_apply(c.array(x)).get()
            assert rel(got, want) < 1e-5
            y = rng.standard_normal(m.ndof)
            xd, yd = c.array(x), c.array(y)
            Mx, My = c.mg_apply(xd, c.empty(m.ndof)), c.mg_apply(yd, c.empty(m.ndof))
            a, b = c.dot(xd, My), c.dot(yd, Mx)
            # (symmetric up to the FP32 roundings of the 16-bit passes' staged vectors)
            assert c.dot(xd, Mx) > 0.0 and c.dot(yd, My) > 0.0 and abs(a - b) <= 1e-6 * np.sqrt(c.dot(xd, Mx) * c.dot(yd, My))
        print("before the solve:", c.precond_info(), c.spmv_sym_info(), "norm f", c.norm(f), flush=True)
        try:
            u, info = c.solve(f, rtol=1e-10)
        except Exception:
            # (round 5: this solve failed inside whole-suite runs only -- say where the first non-finite number is)
            def bad(v):
                a = v.get()
                return "%d of %d entries not finite, first at %s" % ((~np.isfinite(a)).sum(), a.size, np.where(~np.isfinite(a))[0][:5])
            print("f:", bad(f), "| K f:", bad(c.spmv(f, c.empty(m.ndof))), "| K~ f:", bad(c.spmv_lo(f, c.empty(m.ndof))),
                  "| M^-1 f:", bad(c.mg_apply(f, c.empty(m.ndof))), "| diagonal blocks:", bad(c.get_diag_blocks()), flush=True)
            for it in (1, 2, 3, 5, 8):
                try:
                    u2, i2 = c.solve(f, rtol=1e-10, maxit=it)
                except Exception as e:
                    print("maxit", it, "->", str(e)[-160:], flush=True)
            raise
        r = c.axpby(1.0, f, -1.0, c.spmv(u, c.empty(m.ndof)))
        assert c.norm(r) <= 5e-10 * c.norm(f)
        res[mode] = (u.get(), info["iterations"])
        c.close()
    assert rel(res["patch"][0], res["jacobi"][0]) < 1e-7
    assert res["patch"][1] <= 0.9 * res["jacobi"][1], (res["patch"][1], res["jacobi"][1])


def test_dense_inverse_product_does_not_read_behind_its_vector():
    """Round 5 bug: k_gemv_f32x4 (the aggregate level's product where the level has fewer than 2048 unknowns) multiplied the zero padding of
    its rows (lda = a multiple of 32 since round 3) with whatever lay BEHIND the vector -- 0 x NaN = NaN, a solve that broke down only on
    recycled device memory late in long runs.  Here: the vector is the head of a longer buffer whose tail is NaN and infinity."""
    import ctypes as C
    from fcvm_hip._lib import call
    from fcvm_hip.device import Context
    c = Context(0)
    rng = np.random.default_rng(11)
    for n in (385, 7, 64, 100):
        lda = (n + 31) // 32 * 32
        A = np.zeros((n, lda), np.float32)
        A[:, :n] = rng.standard_normal((n, n)).astype(np.float32)
        xb = np.full(n + 40, np.nan)
        xb[n + 1::2] = np.inf
        xb[:n] = rng.standard_normal(n)
        dA = c.array(A.view(np.float32).reshape(-1), dtype=np.float32)
        dx, dy = c.array(xb), c.empty(n)
        call("fcvm_gemv_f32", c.h, C.c_int64(n), C.c_int64(n), C.c_int64(lda), dA.ptr, dx.ptr, dy.ptr)
        got = dy.get()
        want = A[:, :n].astype(np.float64) @ xb[:n]
        assert np.isfinite(got).all() and rel(got, want) < 1e-12, (n, got[:4], want[:4])
    c.close()


def test_level1_fp32_twin_gathers_are_bitwise_the_fp64_gathers(monkeypatch):
    """Round 5: the level-1 passes of the cycle (k_spmv_l1_i16) gather their input from an FP32 twin of the direction vector -- one 16-byte
    record per node, written by whoever writes the vector -- instead of converting the FP64 vector entry by entry: the same roundings, so a
    solve is bitwise the one with FCVM_L1_X32=0.  Single solves and the lockstep pair (second state: its own twins)."""
    from fcvm_hip import examples
    from fcvm_hip.device import Context
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel
    case = examples.plate_with_hole(5.0)
    res = {}
    for x32 in ("1", "0"):
        monkeypatch.setenv("FCVM_L1_X32", x32)
        m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctx=Context(0))
        c = m.ctx
        glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
        f = c.axpby(1.0, c.mul(m.fixdof, glv), 1.0, modf)
        g = c.array(np.random.default_rng(3).standard_normal(m.ndof) * m.fixdof_h)
        u, info = c.solve(f, rtol=1e-10)
        ua, ub, _ia, _ib = c.solve2(f, g, rtol=(1e-10, 1e-9))
        res[x32] = (u.get(), info["iterations"], ua.get(), ub.get())
        assert c.precond_info()["l2_aggregates"] > 0
        c.close()
    assert res["1"][1] == res["0"][1]
    for k in (0, 2, 3):
        assert np.array_equal(res["1"][k], res["0"][k])
    assert np.array_equal(res["1"][0], res["1"][2])        # (and the pair's first vector is the single solve, as before)


def test_level1_boundary_slices_first_is_bitwise_the_unsplit_pass():
    """Multi-GPU layer, round 5: a level-1 smoother pass (k_spmv_l1_i16 with the Chebyshev step as epilogue, rows of shared corner nodes deferred)
    launches the slices with a row of a shared node first (fcvm_mg_l1_set_interface, fcvm_mg_l1_pass_part part 1), so that the exchange of their raw
    products runs beside the other slices (part 2).  Here on one GPU with a made-up interface: after part 1 every deferred raw product is there,
    after part 2 all five vectors are bitwise those of the unsplit pass."""
    import ctypes as C
    from fcvm_hip._lib import call
    case = small_case(12, prescribed=False)
    m = make_model(case)
    c = m.ctx
    m.calc_gsm((0, 0, -9.81))
    nc = c.precond_info()["coarse_nodes"]
    assert nc > 1500
    rng = np.random.default_rng(21)
    shared = np.sort(rng.choice(nc, 60, replace=False)).astype(np.int64)
    defer_h = np.zeros(nc, np.uint8); defer_h[shared] = 1
    defer = c.array(defer_h, dtype=np.uint8)
    din_h, x_h, r_h, t2_h = (rng.standard_normal(3 * nc) for _ in range(4))
    o = (C.c_int64 * 2)()
    call("fcvm_mg_l1_set_interface", c.h, len(shared), shared.ctypes.data, o)
    assert o[0] > 0 and o[1] > 0
    def run(parts):
        din, dout, x, r, t2 = c.array(din_h), c.array(np.full(3 * nc, np.nan)), c.array(x_h), c.array(r_h), c.array(t2_h)
        t_out = c.array(np.full(3 * nc, np.nan))
        snaps = []
        for part in parts:
            call("fcvm_mg_l1_pass_part", c.h, din.ptr, dout.ptr, x.ptr, r.ptr, t2.ptr, t_out.ptr, defer.ptr, C.c_double(0.3), C.c_double(0.7), 1, 1, part)
            snaps.append(t_out.get())
        return [v.get() for v in (dout, x, r, t_out)], snaps
    ref, _ = run([0])
    got, snaps = run([1, 2])
    dofs = (3 * shared[:, None] + np.arange(3)[None, :]).reshape(-1)
    assert np.array_equal(snaps[0][dofs], ref[3][dofs])                        # every deferred raw product is there after part 1
    for a, b in zip(got, ref):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.isnan(ref[0][dofs]).all() and np.isfinite(np.delete(ref[0], dofs)).all()   # (deferred rows are left to fcvm_unpack_sum_cheb)
    # the restriction in the same two parts: the shared corner nodes' entries first (their exchange runs beside the rest)
    r0 = c.array(rng.standard_normal(m.ndof))
    r1_ref, r1 = c.empty(3 * nc), c.array(np.full(3 * nc, np.nan))
    call("fcvm_mg_restrict_part", c.h, r0.ptr, r1_ref.ptr, 0)
    call("fcvm_mg_restrict_part", c.h, r0.ptr, r1.ptr, 1)
    h = r1.get()
    assert np.array_equal(h[dofs], r1_ref.get()[dofs]) and np.isnan(np.delete(h, dofs)).all()
    call("fcvm_mg_restrict_part", c.h, r0.ptr, r1.ptr, 2)
    assert np.array_equal(r1.get(), r1_ref.get())
    call("fcvm_mg_l1_set_interface", c.h, 0, None, None)
