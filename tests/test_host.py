"""CPU-only checks: the C-ABI library loads here and exports every symbol the header declares,
the host-side pattern builder, the text/VTK formats, the mesh generator and the host load vector."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import REPO, load_golden


def test_library_exports_every_declared_symbol():
    from fcvm_hip import _lib
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "fcvm_hip.h")).read()
    names = set(re.findall(r"\b(fcvm_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n
    bound = set(_lib.SIGNATURES) | set(_lib._SPECIAL)
    assert names == bound, names ^ bound
    assert lib.fcvm_version() >= 100


def test_pattern_builder_invariants():
    from fcvm_hip import _lib, meshgen
    for n in (1, 2, 5):
        el, xyz = meshgen.cube_tet10(n, n, n + 1)
        out = (C.c_int64 * 8)()
        _lib.call("fcvm_host_pattern_check", len(el), len(xyz), el.ctypes.data, out)
        r = np.repeat(el - 1, 10, axis=1).reshape(-1)
        c = np.tile(el - 1, (1, 10)).reshape(-1)
        A = sp.coo_matrix((np.ones(len(r)), (r, c)), shape=(len(xyz),) * 2).tocsr()
        assert out[0] == A.nnz            # one block per coupled node pair
        assert out[4] == 0                # element->slot map, columns and row slots consistent
        assert out[1] >= out[0] and out[1] == out[6] * 64
        assert out[3] == np.diff(A.indptr).max()


def test_window_format_invariants():
    """the SpMV's second storage (groups of 512 rows along a Morton curve, LDS windows, prefix storage without padding)
    on a structured and an unstructured mesh: every block once, columns reachable through the window, lanes sorted by length"""
    from fcvm_hip import _lib, examples, meshgen
    el, xyz = meshgen.cube_tet10(9, 8, 10)
    emb = examples.embankment(refine=1)
    for e, x in ((el, xyz), (emb["elNodes"], emb["nocoord"])):
        e, x = np.ascontiguousarray(e, np.int64), np.ascontiguousarray(x, np.float64)
        out = (C.c_int64 * 8)()
        _lib.call("fcvm_host_window_check", len(e), len(x), e.ctypes.data, x.ctypes.data, 0, out)
        assert out[0] == 1 and out[4] == 0, list(out)
        assert out[6] <= out[5] < out[7]                 # stored entries: the blocks + < 16 per wave of alignment, fewer than sliced-ELL slots
        assert out[1] >= -(-len(x) // 512) and 0 < out[3] <= out[2] < 3400
        # a window limit below the largest window: the format is declined (the sliced-ELL kernel stays in charge)
        small = (C.c_int64 * 8)()
        _lib.call("fcvm_host_window_check", len(e), len(x), e.ctypes.data, x.ctypes.data, int(out[2]) - 1, small)
        assert small[0] == 0 and small[2] == out[2]


def test_symmetric_window_form_covers_every_block_once():
    """Host builder of the symmetric form of the window format (fcvm_pattern.cpp build_window_sym): every block of the matrix is
    covered exactly once -- stored, or the transpose of a stored block whose target is the first segment of its row in the same
    group -- the copy kernel's source block is the stored block, lanes stay sorted by length, and a third of the blocks goes."""
    from fcvm_hip import _lib, meshgen
    from conftest import load_golden
    el, xyz = meshgen.cube_tet10(12, 9, 7, 30.0, 20.0, 10.0)
    emb, _ = load_golden("embankment")
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(xyz))                       # a numbering without locality
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    cases = [(el, xyz), (emb["elNodes"], emb["nocoord"]), (inv[el - 1] + 1, xyz[perm])]
    for e, x in cases:
        e, x = np.ascontiguousarray(e, np.int64), np.ascontiguousarray(x, np.float64)
        out = (C.c_int64 * 8)()
        _lib.call("fcvm_host_window_sym_check", len(e), len(x), e.ctypes.data, x.ctypes.data, out)
        if len(x) < 2048 and out[0] == 0:
            continue
        assert out[0] == 1 and out[3] == 0, list(out)
        assert out[2] > 0 and out[6] - out[2] <= out[1] < out[6] - out[2] + 16 * 8 * ((len(x) + 511) // 256)   # entries = kept blocks + alignment
        assert out[1] < 0.85 * out[4]


def test_bad_mesh_is_rejected():
    from fcvm_hip import _lib
    import pytest
    el = np.arange(1, 11, dtype=np.int64).reshape(1, 10)
    el[0, 3] = 99
    out = (C.c_int64 * 8)()
    with pytest.raises(_lib.FcvmError, match="out of range"):
        _lib.call("fcvm_host_pattern_check", 1, 10, el.ctypes.data, out)


def test_cube_mesh_volume_and_orientation(orc):
    from fcvm_hip import meshgen
    el, xyz = meshgen.cube_tet10(3, 2, 4, 3.0, 2.0, 8.0)
    assert len(el) == 6 * 24 and len(xyz) == 7 * 5 * 9
    *_a, V, _lx, _ly, _lz, ne, nn, x = orc.calcGSM(el, xyz, np.array([1.0, 0.3, 1.0]), {}, (0, 0, 0), orc.Loads())
    assert abs(V - 48.0) < 1e-12           # signed volume: every element positively oriented
    # mid-side nodes sit at the midpoints of their corner pairs
    for m, (a, b) in enumerate(meshgen.MID_PAIRS):
        assert np.allclose(xyz[el[:, 4 + m] - 1], 0.5 * (xyz[el[:, a] - 1] + xyz[el[:, b] - 1]))


def test_host_load_vector_matches_oracle(orc):
    from fcvm_hip import loads as L
    from fcvm_hip import meshgen
    case = meshgen.cube_case(3, l=10.0, traction=(1.0, 0.5, 8.0))
    xyz = case["nocoord"]
    rng = np.random.default_rng(0)
    edges = np.stack([case["ufaces"][:5, 0], case["ufaces"][:5, 1], case["ufaces"][:5, 3]], 1)
    kw = dict(pfaces=case["ufaces"][:7], pvals=rng.standard_normal(7), vnodes=np.array([3, 9]),
              vloads=rng.standard_normal((2, 3)), ufaces=case["ufaces"], uloads=case["uloads"], edges=edges,
              eloads=rng.standard_normal((5, 3)))
    for disp in (None, rng.standard_normal(xyz.size) * 0.01):
        a = L.external_loads(xyz, L.Loads(**kw), disp)
        b = orc.external_loads(xyz, orc.Loads(**kw), disp)
        assert np.abs(a - b).max() < 1e-13 * np.abs(b).max()
    tot = L.external_loads(xyz, L.Loads(ufaces=case["ufaces"], uloads=case["uloads"])).reshape(-1, 3).sum(0)
    assert np.allclose(tot, [100.0, 50.0, 800.0])


def test_inp_and_out_formats(tmp_path):
    from fcvm_hip import inpout
    g, meta = load_golden("embankment")
    ctl = inpout.parse_inp_lines(meta["inp"])
    assert ctl["nstep"] == 30 and ctl["gnl"] == "GNLN" and ctl["grav_z"] == -10.0 and ctl["disp_output"] == "incremental"
    p = tmp_path / "x.inp"
    inpout.write_inp(p, ctl)
    assert inpout.read_inp(p) == ctl
    tab = g["out_table"]
    x = np.zeros((int(tab[:, 0].max()) + 1, 3))
    x[tab[:, 0].astype(int)] = tab[:, 1:4]
    out = tmp_path / "x.out"
    inpout.write_out(out, meta["name"], 659, 1418, "GNLN", 30, np.zeros(1), g["out_loadsum"], tab[:, 0].astype(int), x,
                     tab[:, 4], tab[:, 5], tab[:, 6], tab[:, 7], tab[:, 8], tab[:, 9], tab[:, 10], tab[:, 11])
    lines = open(out).read().splitlines()
    assert lines[0] == "model name:" + meta["name"].rjust(50)
    assert lines[3] == "analysis type: elastic-plastic, geometric linear"
    rows = [ln.split() for ln in lines if re.match(r"^\s+\d+\s", ln)]
    assert len(rows) == 31 and np.allclose(np.array(rows, float), tab, rtol=1e-12)


def test_vtk_roundtrip(tmp_path):
    from fcvm_hip import vtkio
    g, _ = load_golden("tensile")
    data = {"Critical Strain Ratio\n": g["vtk_Critical_Strain_Ratio"], "Displacement": g["vtk_Displacement"],
            "Stress Tensor": g["vtk_Stress_Tensor"]}
    f = tmp_path / "t.vtk"
    vtkio.write_vtk(f, g["nocoord"], g["elNodes"] - 1, data)
    pts, cells, back = vtkio.read_vtk(f)
    assert np.array_equal(pts, g["nocoord"]) and np.array_equal(cells, g["elNodes"] - 1)
    assert list(back) == ["Critical_Strain_Ratio\n", "Displacement", "Stress_Tensor"]
    assert np.array_equal(back["Stress_Tensor"], g["vtk_Stress_Tensor"])


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(REPO, "fcvm-workbench_amd")
    for root, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "fcvm_oracle" not in src, f


def test_fcstd_reader_reproduces_reference_mesh_and_first_load_step(orc):
    """tests/golden/tensile.FCStd = Document.xml + FemMesh.unv of the reference's tensile model.
    The reader must give the very mesh the reference solved (cells / nodes of its tensile.vtk, which
    pins the UNV -> fcVM node-order tables), the load sum of its .out header and -- solved with the
    oracle -- the first row and the plastic plateau of its .out table (later rows of that file were
    produced with other step-scaling settings than the shipped .inp and interactive continuation)."""
    from fcvm_hip import fcstd
    m = fcstd.load_model(os.path.join(REPO, "tests", "golden", "tensile.FCStd"))
    g, meta = load_golden("tensile")
    assert np.array_equal(m["elNodes"], g["elNodes"]) and np.array_equal(m["nocoord"], g["nocoord"])
    assert np.allclose(m["material"], [100000.0, 0.0, 7.8e-6])
    assert len(m["fix"]) == 39 and set(m["fix"].values()) == {0.0}
    ctl = orc.read_inp_lines(meta["inp"])
    ctl["target_LF"] = 10.0
    L = m["loads"]
    res = orc.run_analysis(m["elNodes"], m["nocoord"], m["material"], m["fix"], orc.Loads(ufaces=L.ufaces, uloads=L.uloads), ctl)
    assert np.allclose(res["loadsum"], g["out_loadsum"], rtol=1e-9, atol=1e-6)
    tab = g["out_table"]
    mine = np.array([res["lout"], res["un"], res["peeqplot"], res["pplot"], res["svmplot"], res["triaxplot"], res["ecrplot"],
                     res["csrplot"]]).T
    fmt = lambda a: np.array([float("%.2e" % v) for v in a])  # noqa: E731
    assert np.array_equal(fmt(mine[1]), tab[1, 4:12])
    # plateau: load factor 0.5, pressure / von Mises / triaxiality / critical strain as in the reference rows
    assert np.array_equal(fmt(mine[-1][[0, 3, 4, 5, 6]]), tab[-1, [4, 7, 8, 9, 10]])


def test_vm_uniaxial_example_known_answers(orc):
    """BASELINE configs[0]: VM_Uniaxial_Tension_Example (.FCStd + its 17-line .inp of an older macro
    revision: gnl / maxImp / ev1 / ev2 are read as "" = geometrically linear, fcVM.FCMacro:91-94).
    Vertex-referenced supports (statically determinate) and pressure loads on box faces: 10 N/mm^2 pulling on
    the x and z faces, nothing on y.  Closed-form answers the oracle has to hit:
      * elastic, target load factor 3: uniform stress diag(30, 0, 30) in every node, von Mises 30;
      * yield stress lowered to 20: von Mises of the biaxial state is 10 * lambda, so the limit load factor
        of the perfectly plastic cube is exactly 2 (the arc-length steps hover around it within error_max)."""
    from fcvm_hip import fcstd, inpout
    m = fcstd.load_model(os.path.join(REPO, "tests", "golden", "vm_uniaxial.FCStd"))
    meta = json.load(open(os.path.join(REPO, "tests", "golden", "vm_uniaxial.json")))
    ctl = inpout.parse_inp_lines(meta["inp"])
    assert len(meta["inp"]) == 17 and ctl["gnl"] == "" and ctl["nstep"] == 25 and ctl["target_LF"] == 3.0
    assert sorted(m["fix"]) == [3, 4, 5, 9, 11, 17]                      # Vertex2 fixed, Vertex4: x,z  Vertex6: z
    assert np.allclose(m["nocoord"][[1, 3, 5]], [[0, 0, 0], [0, 10, 0], [10, 0, 0]])
    L = m["loads"]
    assert len(L.pfaces) == 24 and sorted(set(np.abs(L.pvals))) == [0.0, 10.0]
    loads = orc.Loads(pfaces=L.pfaces, pvals=L.pvals)
    res = orc.run_analysis(m["elNodes"], m["nocoord"], m["material"], m["fix"], loads, dict(ctl))
    assert res["lout"][-1] == 3.0 and np.abs(res["loadsum"]).max() < 1e-9   # self-equilibrated
    assert np.abs(res["tet10stress"] - [30.0, 0.0, 30.0, 0.0, 0.0, 0.0]).max() < 1e-10
    assert np.abs(res["tet10svm"] - 30.0).max() < 1e-10 and res["peeq"].max() == 0.0
    ctl["sig_yield"] = 20.0
    res = orc.run_analysis(m["elNodes"], m["nocoord"], m["material"], m["fix"], loads, dict(ctl))
    lo = np.asarray(res["lout"])
    assert len(lo) == 26 and abs(lo.max() - 2.0) < 3e-3 and np.all(np.abs(lo[-10:] - 2.0) < 3e-3)
    assert np.abs(res["sigmises"] - 20.0).max() < 1e-9 and res["peeq"].min() > 1e-3


def test_fcstd_reader_rejects_shapes_without_geometry(tmp_path):
    """a constraint on a shape that is neither a plain Part::Box nor backed by a BREP file in the document"""
    import zipfile
    from fcvm_hip import fcstd
    src = zipfile.ZipFile(os.path.join(REPO, "tests", "golden", "tensile.FCStd"))
    xml = src.read("Document.xml").decode("utf8").replace('type="Part::Box"', 'type="Part::Extrusion"')
    p = tmp_path / "x.FCStd"
    with zipfile.ZipFile(p, "w") as z:
        z.writestr("Document.xml", xml)
        z.writestr("FemMesh.unv", src.read("FemMesh.unv"))
    with pytest.raises(NotImplementedError):
        fcstd.load_model(str(p))


def test_brep_reader_on_reference_shapes():
    """fcvm_hip.brep on the geometry files of two reference documents.  Embankment (Part::Extrusion: 12 faces from
    11 face TShapes -- the two caps share one): the faces its constraints reference select exactly the dofs that
    are exactly zero in the reference's own displacement result (tests/golden/embankment.npz).  Plate_with_hole
    (Part::Cut of a box and a cylinder): areas / lengths against closed forms, and the node sets
    examples.plate_with_hole constrains are those of Face1 / Face7 / Vertex1,2,5 of the document."""
    import zipfile
    from fcvm_hip import brep, examples, fcstd
    g, _ = load_golden("embankment")
    m = fcstd.load_model(os.path.join(REPO, "tests", "golden", "embankment.FCStd"))
    assert np.array_equal(m["elNodes"], g["elNodes"]) and np.array_equal(m["nocoord"], g["nocoord"])
    assert sorted(m["fix"]) == g["fixed_dofs"].tolist() and set(m["fix"].values()) == {0.0}
    assert np.allclose(m["material"], g["material"])
    z = zipfile.ZipFile(os.path.join(REPO, "tests", "golden", "embankment.FCStd"))
    b = brep.BrepShape(z.read("Extrude.Shape.brp").decode())
    assert (len(b.faces), len(b.edges), len(b.vertices)) == (12, 30, 20)
    assert b.face(11).area == pytest.approx(5.3e7) and b.face(12).surface[1][1] == pytest.approx(-1000.0)
    z = zipfile.ZipFile(os.path.join(REPO, "tests", "golden", "plate_with_hole.FCStd"))
    b = brep.BrepShape(z.read("PartShape2.brp").decode())
    assert (len(b.faces), len(b.edges), len(b.vertices)) == (7, 15, 10)
    assert b.face(2).surface[0] == "cylinder" and b.face(2).area == pytest.approx(np.pi * 50.0 / 2.0 * 10.0, rel=1e-9)
    assert b.face(3).area == pytest.approx(200.0 * 100.0 - np.pi * 2500.0 / 4.0, rel=1e-5)
    assert sorted(round(b.edge(i).length, 6) for i in range(1, 16))[-1] == 200.0
    assert b.edge(5).length == pytest.approx(np.pi * 25.0, rel=1e-5)
    case = examples.plate_with_hole(10.0)
    xyz = case["nocoord"]
    fix = {}
    for n in np.where(b.face(1).nodes(xyz, 1e-6))[0]:
        fix[3 * int(n)] = 0.0
    for n in np.where(b.face(7).nodes(xyz, 1e-6))[0]:
        fix[3 * int(n) + 1] = 0.0
    for v in (1, 2, 5):
        (n,) = np.where(b.vertex(v).nodes(xyz, 1e-6))[0]
        fix[3 * int(n) + 2] = 0.0
    assert fix == case["fix"]
    on_hole = b.face(2).nodes(xyz, 1e-6)
    assert on_hole.sum() > 50 and np.allclose(np.hypot(xyz[on_hole, 0], xyz[on_hole, 1]), 50.0)
    loaded = b.face(7).nodes(xyz, 1e-6)
    assert np.array_equal(np.unique(case["ufaces"]), np.where(loaded)[0] + 1)


def test_embankment_document_to_out_file_is_the_reference_file(orc, tmp_path):
    """The whole caller chain on the CPU side: the reference's FreeCAD document (mesh + BREP + constraints) and its
    control file through fcstd.load_model -> oracle analysis -> inpout.write_out give the reference's shipped .out
    file byte for byte."""
    from fcvm_hip import fcstd, inpout
    g, meta = load_golden("embankment")
    m = fcstd.load_model(os.path.join(REPO, "tests", "golden", "embankment.FCStd"))
    ctl = inpout.parse_inp_lines(meta["inp"])
    res = orc.run_analysis(m["elNodes"], m["nocoord"], m["material"], m["fix"], orc.Loads(), dict(ctl))
    out = tmp_path / "e.out"
    inpout.write_out(out, meta["name"], len(m["elNodes"]), len(m["nocoord"]), ctl["gnl"], ctl["nstep"], res["eigenval"], res["loadsum"],
                     res["crip"], res["x"], res["lout"], res["un"], res["peeqplot"], res["pplot"], res["svmplot"], res["triaxplot"],
                     res["ecrplot"], res["csrplot"])
    assert open(out).read() == open(os.path.join(REPO, "tests", "golden", "embankment_reference.out")).read()


def test_simple_shear_document_known_answer(orc):
    """The reference's `Simple Shear` document (cube under self-equilibrated shear tractions of 100 N/mm^2 on four faces,
    ConstraintForce on faces, vertex supports) with its control file: von Mises yields in pure shear at
    tau = sig_yield / sqrt(3), so the perfectly plastic limit load factor is 0.57735."""
    from fcvm_hip import fcstd, inpout
    docs = json.load(open(os.path.join(REPO, "tests", "golden", "documents.json")))
    m = fcstd.load_model(os.path.join(REPO, "tests", "golden", "simple_shear.FCStd"))
    ctl = inpout.parse_inp_lines(docs["simple_shear"]["inp"])
    assert ctl["sig_yield"] == 100.0 and ctl["Et_E"] == 0.0
    L = m["loads"]
    assert len(L.ufaces) == 16 and np.allclose(np.abs(L.uloads).max(1), 100.0)
    res = orc.run_analysis(m["elNodes"], m["nocoord"], m["material"], m["fix"], orc.Loads(ufaces=L.ufaces, uloads=L.uloads), dict(ctl))
    assert np.abs(res["loadsum"]).max() < 1e-9
    plateau = np.asarray(res["lout"])[-12:]     # the control file's error_max is 1 %: the steps scatter by that much
    assert np.abs(plateau - 1.0 / np.sqrt(3.0)).max() < 0.012 and abs(plateau.mean() - 1.0 / np.sqrt(3.0)) < 3e-3
    assert np.abs(res["sigmises"] - 100.0).max() < 1e-6


def test_calcSum_and_avr_format(tmp_path):
    """fcVM.calcSum / .avr writer (host post-processing): exact length / area of a box edge / face,
    averages of linear nodal fields, and the file layout of the reference's shipped .avr files."""
    import fcVM as ft
    from fcvm_hip import meshgen
    el, xyz = meshgen.cube_tet10(2, 2, 2, 4.0, 6.0, 10.0)
    faces = meshgen.boundary_faces_on_plane(el, xyz, 2, 10.0)                       # top face, area 24
    edge = [[f[0], f[1], f[3]] for f in faces if abs(xyz[f[0] - 1][1]) < 1e-9 and abs(xyz[f[1] - 1][1]) < 1e-9]
    assert len(edge) == 2                                                          # the edge y = 0, z = 10: length 4
    field = 2.0 + 3.0 * xyz[:, 0]                                                  # linear in x
    out = ft.calcSum([edge], [faces.tolist()], xyz, field, 2 * field, np.ones(len(xyz)))
    el_, ep, ec, es, fa, fp, fc, fs = out
    assert el_[0] == pytest.approx(4.0) and fa[0] == pytest.approx(24.0)
    assert ec[0] == pytest.approx(2.0 + 3.0 * 2.0) and ep[0] == pytest.approx(2 * 8.0) and es[0] == pytest.approx(1.0)
    assert fc[0] == pytest.approx(8.0) and fs[0] == pytest.approx(1.0)
    p = tmp_path / "m.avr"
    ft.write_avr(p, "m", ["Box(Cube).Edge1"], ["Box(Cube).Face6"], *out)
    lines = open(p).read().splitlines()
    assert lines[0] == "model name:" + "m".rjust(30) and lines[2] == "average values"
    assert lines[5].split() == ["Length", "peeq", "CSR", "svmises", "edge"]
    assert lines[6].split() == ["4.00e+00", "1.60e+01", "8.00e+00", "1.00e+00", "Box(Cube).Edge1"]
    assert lines[10].split() == ["Area", "peeq", "CSR", "svmises", "face"]


def test_spatial_element_order_bounds_the_interfaces_of_a_shuffled_mesh():
    """partition_model's default element order (Morton curve through the centroids): on the Embankment mesh refined twice
    (42k tet10) with SHUFFLED element numbering an 8-way range partition keeps its interfaces small, where ranges of the
    shuffled list share nearly every node; on the unshuffled list it is no worse than the mesh generator's own locality."""
    from fcvm_hip import examples
    from fcvm_hip.dist import interface_node_counts, spatial_element_order
    case = examples.embankment(refine=2)
    el, xyz = case["elNodes"], case["nocoord"]
    perm = np.random.default_rng(7).permutation(len(el))
    shuffled = el[perm]
    order = spatial_element_order(shuffled, xyz)
    assert np.array_equal(np.sort(order), np.arange(len(el)))
    shared_s, held_s = interface_node_counts(shuffled, order, 8)
    shared_r, held_r = interface_node_counts(shuffled, None, 8)
    shared_g, held_g = interface_node_counts(el, None, 8)
    shared_o, held_o = interface_node_counts(el, spatial_element_order(el, xyz), 8)
    assert min(s / h for s, h in zip(shared_r, held_r)) > 0.95          # ranges of a shuffled list: everything is interface
    assert max(s / h for s, h in zip(shared_s, held_s)) < 0.30          # Morton ranges: a surface-to-volume share
    assert sum(shared_s) == sum(shared_o) and sum(shared_o) < 1.3 * sum(shared_g)
    assert max(held_s) < 1.25 * len(xyz) / 8 + max(shared_s)


def _bipyramid_fan(k):
    """2k tet10 around one hub node: ring of k points, two apexes; the hub couples to 5k + 4 nodes"""
    import itertools
    ang = 2 * np.pi * np.arange(k) / k
    corners = [(0.0, 0.0, 0.0)] + [(np.cos(a), np.sin(a), 0.0) for a in ang] + [(0.0, 0.0, 1.0), (0.0, 0.0, -1.0)]
    top, bot = k + 1, k + 2
    tets = []
    for i in range(k):
        a, b = 1 + i, 1 + (i + 1) % k
        tets.append((0, a, b, top))
        tets.append((0, b, a, bot))
    xyz = [np.array(c) for c in corners]
    mid = {}
    el = []
    pairs = [(0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)]
    for t in tets:
        row = list(t)
        for a, b in pairs:
            key = (min(t[a], t[b]), max(t[a], t[b]))
            if key not in mid:
                mid[key] = len(xyz)
                xyz.append(0.5 * (xyz[key[0]] + xyz[key[1]]))
            row.append(mid[key])
        el.append(row)
    return np.array(el, np.int64) + 1, np.array(xyz, np.float64)


def test_symmetric_window_form_is_declined_where_a_row_would_receive_too_many_transposed_products():
    """the 64-bit fixed-point sums of k_spmv_win_sym hold 85 transposed products per row without wrap-around
    (fcvm_pattern.h SYM_MAX_RECEIVED); build_window_sym counts what every row receives and declines the form beyond that --
    a hub node with 244 neighbours in one group receives about half of them -- and check_window_sym recounts it."""
    from fcvm_hip import _lib
    for k, built in ((20, 1), (48, 0)):
        e, x = _bipyramid_fan(k)
        assert np.bincount(e.ravel()).max() == 2 * k                   # the hub sits in every element
        out = (C.c_int64 * 8)()
        _lib.call("fcvm_host_window_sym_check", len(e), len(x), e.ctypes.data, x.ctypes.data, out)
        assert out[0] == built, (k, list(out))
        if built:
            assert out[3] == 0 and out[2] > 0
