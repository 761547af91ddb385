"""The CPU oracle against the reference's own result files (tests/golden, made by
tests/golden/make_golden.py from /root/reference's output / control / FreeCAD data files)."""
import numpy as np

from conftest import load_golden


def _run(orc, g, meta):
    ctl = orc.read_inp_lines(meta["inp"])
    fix = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    return ctl, orc.run_analysis(g["elNodes"], g["nocoord"], g["material"], fix, orc.Loads(), ctl)


def sig3(a, b):
    """agreement to the 3 significant digits the reference prints ('%.2e')"""
    fa = np.array([float("%.2e" % v) for v in np.asarray(a).ravel()])
    return np.allclose(fa, np.asarray(b).ravel(), rtol=1.1e-2, atol=0)


def test_embankment_out_table(orc, embankment):
    """30-step elastoplastic collapse analysis: every column of the reference's .out table."""
    g, meta = embankment
    ctl, res = _run(orc, g, meta)
    tab = g["out_table"]
    assert len(res["lout"]) == len(tab) == ctl["nstep"] + 1
    assert np.array_equal(res["crip"], tab[:, 0].astype(int))
    x = res["x"][res["crip"]]
    for col, val in ((1, x[:, 0]), (2, x[:, 1]), (3, x[:, 2]), (4, res["lout"]), (5, res["un"]), (6, res["peeqplot"]),
                     (7, res["pplot"]), (8, res["svmplot"]), (9, res["triaxplot"]), (10, res["ecrplot"]),
                     (11, res["csrplot"])):
        assert sig3(val, tab[:, col]), "column %d" % col
    assert sig3(res["loadsum"], g["out_loadsum"])


def test_embankment_vtk_fields(orc, embankment):
    """final nodal fields of the reference's .vtk, stored in f64"""
    g, meta = embankment
    ctl, res = _run(orc, g, meta)
    rel = lambda a, b: np.abs(np.asarray(a) - b).max() / np.abs(b).max()  # noqa: E731
    assert rel(res["displacements"], g["vtk_Displacement"].reshape(-1)) < 1e-9
    assert rel(res["tet10stress"], g["vtk_Stress_Tensor"]) < 1e-9
    assert rel(res["tet10peeq"], g["vtk_Equivalent_Plastic_Strain"]) < 1e-9
    assert rel(res["tet10csr"], g["vtk_Critical_Strain_Ratio"]) < 1e-9
    assert rel(res["tet10svm"], g["vtk_von_Mises_Stress"]) < 1e-9
    assert rel(res["tet10triax"], g["vtk_Triaxiality"]) < 1e-9


def test_principal_stress_against_reference_vtk(orc):
    for name in ("embankment", "lug", "tensile", "buckling"):
        g, _ = load_golden(name)
        s1, s2, s3, v1, v2, v3 = orc.calculate_principal_stress(g["vtk_Stress_Tensor"])
        scale = np.abs(g["vtk_Stress_Tensor"]).max()
        assert np.abs(s1 - g["vtk_Major_Principal_Stress"]).max() <= 1e-10 * scale
        assert np.abs(s2 - g["vtk_Intermediate_Principal_Stress"]).max() <= 1e-10 * scale
        assert np.abs(s3 - g["vtk_Minor_Principal_Stress"]).max() <= 1e-10 * scale
        # vectors: eigenvalue * unit eigenvector, sign free; compare where the eigenvalue is simple
        gap = np.minimum(np.abs(s1 - s2), np.abs(s2 - s3)) > 1e-6 * scale
        for mine, ref in ((v1, g["vtk_Major_Principal_Stress_Vector"]), (v3, g["vtk_Minor_Principal_Stress_Vector"])):
            d = np.minimum(np.abs(mine - ref).max(1), np.abs(mine + ref).max(1))
            assert (not gap.any()) or d[gap].max() < 1e-6 * scale


def lug_pressure_case(g, meta, orc_or_none=None):
    """Lug_Pressure_Example from the fixture: (fix, pressure faces, pressure values, control dict lines)."""
    fix = {int(d): 0.0 for d in g["fixed_dofs"]}
    return fix, g["pfaces"], g["pvals"], meta["inp"]


def rel_each(a, b, floor=1e-3):
    """largest ELEMENT-WISE relative difference over the entries of b above floor * max |b|"""
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    m = np.abs(b) > floor * np.abs(b).max()
    return float((np.abs(a - b)[m] / np.abs(b)[m]).max()) if m.any() else 0.0


def check_lug_against_reference_files(res, g, tol, tol_each=None):
    """all 61 rows of the reference's .out (two passes of 30 steps), every column, and the f64 fields of its .vtk (tol: relative to the
    field maximum; tol_each: entry by entry over the entries above 1e-3 of it)"""
    tab = g["out_table"]
    assert len(res["lout"]) == len(tab) == 61
    assert np.array_equal(np.asarray(res["crip"]), tab[:, 0].astype(int))
    x = res["x"][np.asarray(res["crip"])]
    for col, val in ((1, x[:, 0]), (2, x[:, 1]), (3, x[:, 2]), (4, res["lout"]), (5, res["un"]), (6, res["peeqplot"]),
                     (7, res["pplot"]), (8, res["svmplot"]), (9, res["triaxplot"]), (10, res["ecrplot"]), (11, res["csrplot"])):
        assert sig3(val, tab[:, col]), "column %d" % col
    assert sig3(res["loadsum"][0], g["out_loadsum"][0]) and abs(res["loadsum"][1]) < 1e-6 and abs(res["loadsum"][2]) < 1e-6
    rel = lambda a, b: np.abs(np.asarray(a) - b).max() / np.abs(b).max()  # noqa: E731
    # disp_output = incremental (fcVM.py:1631-1634): the file holds the displacement increment of the last step
    assert rel(np.asarray(res["displacements"]).reshape(-1, 3), g["vtk_Displacement"]) < tol
    assert rel(res["tet10stress"], g["vtk_Stress_Tensor"]) < tol
    assert rel(res["tet10peeq"], g["vtk_Equivalent_Plastic_Strain"]) < tol
    assert rel(res["tet10csr"], g["vtk_Critical_Strain_Ratio"]) < tol
    assert rel(res["tet10svm"], g["vtk_von_Mises_Stress"]) < tol
    assert rel(res["tet10triax"], g["vtk_Triaxiality"]) < tol
    if tol_each is not None:
        each = {k: rel_each(a, b) for k, a, b in (("displacement", np.asarray(res["displacements"]).reshape(-1, 3), g["vtk_Displacement"]),
                                                   ("stress", res["tet10stress"], g["vtk_Stress_Tensor"]),
                                                   ("von Mises", res["tet10svm"], g["vtk_von_Mises_Stress"]),
                                                   ("peeq", res["tet10peeq"], g["vtk_Equivalent_Plastic_Strain"]))}
        assert max(each.values()) < tol_each, each


def test_lug_pressure_example_out_and_vtk(orc):
    """Pressure loads on curved faces (face integrals with the element normal, fcVM.py:649-672), `incremental`
    displacement output and the continuation pass after the plot window (fcVM.py:1304-1308, 1615), pinned on the
    reference's own Lug_Pressure_Example.out / .vtk: 100 N/mm2 on the two half-cylinders of the hole, 2 x 30 steps."""
    g, meta = load_golden("lug")
    fix, pf, pv, inp = lug_pressure_case(g, meta)
    ctl = orc.read_inp_lines(inp)
    assert ctl["disp_output"] == "incremental" and ctl["nstep"] == 30
    res = orc.run_analysis(g["elNodes"], g["nocoord"], g["material"], fix, orc.Loads(pfaces=pf, pvals=pv), ctl, max_passes=2)
    check_lug_against_reference_files(res, g, 1e-9)
    # one pass only = the first 31 rows
    one = orc.run_analysis(g["elNodes"], g["nocoord"], g["material"], fix, orc.Loads(pfaces=pf, pvals=pv), ctl)
    assert len(one["lout"]) == 31 and np.array_equal(one["lout"], res["lout"][:31])


def _bar(orc):
    g, meta = load_golden("barbuckling")
    fix = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    return g, meta, fix, orc.Loads(ufaces=g["ufaces"], uloads=g["uloads"])


def test_bar_buckling_factors(orc):
    """Reference 'Bar buckling' model (GNLY): the two elastic buckling factors of its .out header
    (fcVM.py:1199-1214: K and geometric stiffness of calcTSM(nstep = 1), ARPACK buckling mode)."""
    g, meta, fix, loads = _bar(orc)
    w, v = orc.buckling_analysis(g["elNodes"], g["nocoord0"], g["material"], fix, (0.0, 0.0, 0.0), loads)
    assert np.allclose(np.sort(w), g["buckling_factors"], rtol=0, atol=6e-6)   # printed with 5 decimals
    assert abs(np.sort(w)[0] - np.pi ** 2 * g["material"][0] * (10.0 ** 4 / 12) / (4 * 1000.0 ** 2) / 1000.0) < 0.01  # Euler


def test_bar_buckling_gnly_analysis(orc):
    """The geometrically non-linear elastic-plastic run of the same model: updated-Lagrange tangent
    re-assembly (calcTSM), stress convection and Riks control, started from the imperfect geometry the
    reference exported.  Every row of its .out table and its final f64 nodal fields."""
    g, meta, fix, loads = _bar(orc)
    ctl = orc.read_inp_lines(meta["inp"])
    assert ctl["gnl"] == "GNLY" and float(ctl["maxImp"]) == 10.0
    res = orc.run_analysis(g["elNodes"], g["nocoord"], g["material"], fix, loads, ctl)
    tab = g["out_table"]
    assert len(res["lout"]) == len(tab) == 11
    for col, val in ((4, res["lout"]), (5, res["un"]), (6, res["peeqplot"]), (7, res["pplot"]), (8, res["svmplot"]),
                     (9, res["triaxplot"]), (10, res["ecrplot"]), (11, res["csrplot"])):
        assert sig3(val, tab[:, col]), "column %d" % col
    rel = lambda a, b: np.abs(np.asarray(a) - b).max() / np.abs(b).max()  # noqa: E731
    # 1e-7: the exported coordinates carry the reference's imperfection on the (nominally) clamped nodes too (1.6e-6 mm)
    assert rel(res["disp_el"], g["vtk_Elastic_Displacement"].reshape(-1)) < 1e-7
    assert rel(res["displacements"], g["vtk_Displacement"].reshape(-1)) < 1e-7
    assert rel(res["tet10stress"], g["vtk_Stress_Tensor"]) < 1e-7
    assert rel(res["tet10svm"], g["vtk_von_Mises_Stress"]) < 1e-7


def test_buckling_model_equilibrium_and_factors(orc):
    """Reference 'Buckling' model (linear buckling run, nstep = 1, ConstraintDisplacement + end block).
    (1) K u_ref = f: the reference's own f64 elastic displacement field is in equilibrium with the
        oracle's stiffness and its consistent face load -- residual 1e-10 of the load at every free dof;
    (2) the elastic solution itself; (3) the buckling factors / shapes of its .out / .vtk.  Those were
        written by an earlier revision of the reference (shapes scaled to max 1, mesh older than the one in
        the .FCStd): factors agree to 5e-6 relative, shapes to 2e-6 of their maximum."""
    import scipy.sparse as sp
    g, meta = load_golden("buckling")
    fix = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    loads = orc.Loads(ufaces=g["ufaces"], uloads=g["uloads"])
    el, xyz, mat = g["elNodes"], g["nocoord"], g["material"]
    nn = len(xyz)
    stm, row, col, glv, *_ = orc.calcGSM(el, xyz, mat, {}, (0.0, 0.0, 0.0), loads)
    K = sp.coo_matrix((stm, (row, col)), shape=(3 * nn, 3 * nn)).tocsr()
    K = K + sp.tril(K, -1).T
    u_ref = g["vtk_Elastic_Displacement"].reshape(-1)
    free = np.ones(3 * nn, bool)
    free[g["fixed_dofs"]] = False
    assert np.abs((K @ u_ref - glv)[free]).max() < 1e-10 * np.abs(glv).max()
    assert abs(glv.reshape(-1, 3)[:, 0].sum() - g["out_loadsum"][0]) < 1e-9
    res = orc.run_analysis(el, xyz, mat, fix, loads, orc.read_inp_lines(meta["inp"]))
    assert np.abs(res["disp_el"] - u_ref).max() < 1e-7 * np.abs(u_ref).max()  # slender bar: cond(K) ~ 1e9
    assert np.array_equal(res["lout"], [0.0]) and len(g["out_table"]) == 1  # fcVM.py:1193
    w, v = res["eigenval"], res["eigenvec"]
    assert np.allclose(w, g["buckling_factors"], rtol=5e-6, atol=0)
    for i, key in enumerate(("vtk_Buckling_shape_for_lambda1__4263", "vtk_Buckling_shape_for_lambda2__38524")):
        r, x = g[key].reshape(-1), v[:, i]
        s = (x @ r) / (x @ x)
        assert np.abs(s * x - r).max() < 5e-6   # shapes ~0.5; the stresses behind K_g come from a solve with cond(K) ~ 1e9


def test_plate_with_hole_collapse_curve(orc):
    """BASELINE configs[1]: the reference's Plate_with_hole_Example.  Its document has geometry, constraints and
    loads but no mesh, so the oracle runs the shipped control file on the mapped mesh of
    fcvm_hip.meshgen.plate_with_hole (480 tet10 at h = 25 against the reference's 395 gmsh elements): different
    meshes of the same problem, so the comparison is physical -- every row of the reference's load / displacement
    load history within 2 % (displacements of the first four steps too), the same plateau near 0.23 (tension + bending of the eccentric ligament; beam
    theory: 4 l + 4 l^2 = 1 -> 0.207 without the constraint of the third stress component)."""
    import json
    import os
    from conftest import REPO
    from fcvm_hip import examples
    ref = json.load(open(os.path.join(REPO, "tests", "golden", "plate_with_hole.json")))
    m = examples.plate_with_hole(25.0)
    assert list(examples.PLATE_WITH_HOLE_INP) == ref["inp"] and np.allclose(m["material"], ref["material"])
    res = orc.run_analysis(m["elNodes"], m["nocoord"], m["material"], m["fix"], orc.Loads(ufaces=m["ufaces"], uloads=m["uloads"]),
                           orc.read_inp_lines(ref["inp"]))
    tab = np.array(ref["out_table"])
    assert np.allclose(res["loadsum"], ref["loadsum"], atol=1e-6)
    assert len(res["lout"]) == len(tab) == 11
    assert np.abs(res["lout"][1:] / tab[1:, 4] - 1.0).max() < 0.02
    assert np.abs(res["un"][1:5] / tab[1:5, 5] - 1.0).max() < 0.02  # later step lengths follow the iteration counts
    assert 0.207 < res["lout"][-1] < 0.24


def test_u100_column_buckling_and_gnly_run(orc):
    """Reference 'buckling_U100_3D' model (GNLY): extruded U-profile column, clamped at z = 0, 50 kN axial + 2 kN lateral
    on the end face, gravity, imperfection 1 mm along the first buckling shape.  From the PERFECT geometry
    (make_golden.u100): the two buckling factors of the .out header (5 decimals), both buckling shapes of the .vtk
    (1e-7 of their maximum), the imperfect coordinates the reference wrote (1e-6 mm), its load sums and every row of the
    load / displacement table: all three passes of the reference's session (multi-pass continuation with new targets)."""
    g, meta = load_golden("u100")
    fix = {int(d): float(v) for d, v in zip(g["fixed_dofs"], g["fixed_vals"])}
    loads = orc.Loads(ufaces=g["ufaces"], uloads=g["uloads"])
    ctl = orc.read_inp_lines(meta["inp"])
    assert ctl["gnl"] == "GNLY" and ctl["grav_z"] == -10.0 and ctl["csr_option"] == "CSR"
    # three passes like the reference's session: target 0.8 from the control file, then 1.0 typed into the plot window
    # ("add": restart with 1 / nstep of the elastic solution, fcVM.py:1664-1672), then a target beyond the peak load
    res = orc.run_analysis(g["elNodes"], g["nocoord0"], g["material"], fix, loads, ctl, use_imperfection=True, max_passes=3,
                           pass_targets=[1.0, 1000.0])
    assert np.allclose(res["eigenval"], g["buckling_factors"], rtol=0, atol=6e-6)
    for i, key in enumerate(("shape1", "shape2")):
        r, x = g[key].reshape(-1), res["eigenvec"][:, i]
        assert np.abs((x @ r) / (x @ x) * x - r).max() < 1e-7
    assert np.abs(res["nocoord"] - g["nocoord"]).max() < 1e-6
    assert np.allclose(res["loadsum"], g["out_loadsum"], rtol=2e-3, atol=1e-9)     # printed with 3 digits
    assert np.abs(res["disp_el"] - g["elastic_displacement"].reshape(-1)).max() < 1e-6 * np.abs(g["elastic_displacement"]).max()
    tab = g["out_table_full"]    # all 22 rows: 8 steps to 0.8, 3 steps to 1.0, 10 steps over the peak (1.19) down to 1.13
    assert len(res["lout"]) == len(tab) == 22 and res["lout"][8] == 0.8 and res["lout"][11] == 1.0
    assert np.array_equal(res["crip"], tab[:, 0].astype(int))
    for col, val in ((4, res["lout"]), (5, res["un"]), (6, res["peeqplot"]), (7, res["pplot"]), (8, res["svmplot"]),
                     (9, res["triaxplot"]), (10, res["ecrplot"]), (11, res["csrplot"])):
        assert sig3(val, tab[:, col]), "column %d" % col
    # final nodal fields of the .vtk (written after the third pass) through calcSum (fcVM.py:2830-2900) over the selection
    # of the shipped .avr summary (Edge7 / Face2 of the Extrude): the oracle's fields average to what the reference's do
    import fcVM as ft
    sel = ([g["avr_edge7"].tolist()], [g["avr_face2"].tolist()], g["nocoord0"])
    want = ft.calcSum(*sel, g["vtk_csr"], g["vtk_peeq"], g["vtk_svm"])
    got = ft.calcSum(*sel, res["tet10csr"], res["tet10peeq"], res["tet10svm"])
    for w, v in zip(want, got):
        assert abs(v[0] - w[0]) <= 1e-6 * max(abs(w[0]), 1e-3), (w, v)


def test_calcSum_on_the_reference_avr_selection(tmp_path):
    """`output files/buckling_U100_3D.avr` (fcVM_sum.FCMacro:110-160): calcSum over the 3-node line elements on CAD Edge7 and the
    6-node triangles on CAD Face2 of the reference's own mesh gives the file's Length 5.00e+01 and Area 5.00e+04, and
    write_avr reproduces its layout line for line.  The three averages of each row cannot be pinned: the shipped .avr was
    written from another state of the model than the shipped .vtk of the same name -- that .vtk has peeq = 0 on all of Edge7
    (.avr: 9.66e-05) and no edge or face of the shape averages to the .avr's numbers (scanned in round 3) -- so for them
    the .vtk's nodal fields are the pinned data (test_u100_column_buckling_and_gnly_run and its GPU twin)."""
    import fcVM as ft
    g, meta = load_golden("u100")
    ref = meta["avr"]
    out = ft.calcSum([g["avr_edge7"].tolist()], [g["avr_face2"].tolist()], g["nocoord0"], g["vtk_csr"], g["vtk_peeq"], g["vtk_svm"])
    assert len(g["avr_edge7"]) == 1 and len(g["avr_face2"]) == 40
    p = tmp_path / "u100.avr"
    ft.write_avr(p, meta["name"], ["Extrude(Extrude).Edge7"], ["Extrude(Extrude).Face2"], *out)
    mine = open(p).read().splitlines()
    assert len(mine) == len(ref)
    for i, (a, b) in enumerate(zip(mine, ref)):
        if i in (6, 11):     # the two data rows: size and name columns
            assert a.split()[0] == b.split()[0] and a.split()[-1] == b.split()[-1] and len(a) == len(b)
        else:
            assert a == b, (i, a, b)
    assert ["%.2e" % out[0][0], "%.2e" % out[4][0]] == ["5.00e+01", "5.00e+04"]
    # averages of the shipped .vtk state over the same selection (what the GPU run is compared with)
    assert ["%.2e" % v[0] for v in out[1:4]] == ["0.00e+00", "0.00e+00", "2.09e+02"]
    assert ["%.2e" % v[0] for v in out[5:]] == ["2.96e-06", "1.48e-05", "1.02e+02"]
