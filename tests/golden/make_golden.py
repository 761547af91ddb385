#!/usr/bin/env python3
"""Builds the golden fixtures under tests/golden/ from the reference's own DATA files.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py

What is read (data files, no reference source is imported, executed or copied):
  output files/<model>.vtk   mesh (points, tet10 cells in fcVM's internal node order: exportVTK,
                             fcVM.py:2907-2912 writes elNodes-1 verbatim) and the final nodal
                             result fields in f64
  output files/<model>.out   the load / displacement / peeq / csr table (3 significant digits)
  control files/<model>.inp  the 21-line analysis control file
  freeCAD files/<model>.FCStd (zip) Document.xml: material card and the constraint objects

The FreeCAD geometry kernel is not available, so the node sets of the constrained CAD faces are
resolved geometrically (faces of the Embankment extrusion are the planes of its bounding box) and
cross-checked against the exact-zero pattern of the reference's own displacement result: both must
give the same sets, otherwise the script aborts.
"""
import json
import os
import re
import shutil
import sys
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip.vtkio import read_vtk  # noqa: E402

REF = "/root/reference"


def parse_out(path):
    rows, head = [], {}
    for ln in open(path):
        if ln.startswith("No. of elements"):
            head["ne"] = int(ln.split(":")[1])
        elif ln.startswith("No. of Degrees"):
            head["nn"] = int(ln.split(":")[1])
        elif ln.startswith("Sum of loads"):
            head.setdefault("loadsum", []).append(float(ln.split(":")[1]))
        else:
            t = ln.split()
            if len(t) == 12 and re.match(r"^-?\d+$", t[0]):
                rows.append([float(v) for v in t])
    return head, np.array(rows)


def material_from_fcstd(path):
    xml = zipfile.ZipFile(path).read("Document.xml").decode("utf8", "replace")

    def item(key):
        return re.search(r'<Item key="%s" value="([^"]*)"' % key, xml).group(1)

    def qty(s, units):
        v, u = s.split()
        return float(v) * units[u]

    E = qty(item("YoungsModulus"), {"MPa": 1.0, "kPa": 1e-3, "GPa": 1e3, "Pa": 1e-6})
    nu = float(item("PoissonRatio"))
    rho = qty(item("Density"), {"kg/m^3": 1e-9, "kg/mm^3": 1.0})
    return E, nu, rho, xml


def embankment():
    name = "Embankment_with_Ditch_Example"
    pts, cells, data = read_vtk(os.path.join(REF, "output files", name + ".vtk"))
    head, table = parse_out(os.path.join(REF, "output files", name + ".out"))
    inp = [ln.strip() for ln in open(os.path.join(REF, "control files", name + ".inp"))]
    inp = [s for s in inp if s != ""][:21]
    E, nu, rho, xml = material_from_fcstd(os.path.join(REF, "freeCAD files", name + ".FCStd"))
    assert head["ne"] == len(cells) and head["nn"] == len(pts)
    # constraints in the FreeCAD document: ConstraintFixed on 3 faces (bottom + the two x-ends),
    # ConstraintDisplacement (yFree = false, value 0) on the two extrusion caps (plane strain)
    assert xml.count('<Object type="Fem::ConstraintFixed"') == 1
    assert xml.count('<Object type="Fem::ConstraintDisplacement"') == 1
    lo, hi = pts.min(0), pts.max(0)
    tol = 1e-6
    on = lambda ax, v: np.abs(pts[:, ax] - v) < tol  # noqa: E731
    fixed_nodes = np.where(on(0, lo[0]) | on(0, hi[0]) | on(2, lo[2]))[0]
    ycap_nodes = np.where(on(1, lo[1]) | on(1, hi[1]))[0]
    fix = {}
    for n in fixed_nodes:
        for k in range(3):
            fix[3 * int(n) + k] = 0.0
    for n in ycap_nodes:
        fix[3 * int(n) + 1] = 0.0
    # cross-check with the reference's own result: fixed dofs are exactly the dofs whose
    # displacement is exactly 0.0 in the exported field
    u = data["Displacement"].reshape(-1)
    assert set(np.where(u == 0.0)[0].tolist()) == set(fix.keys()), "BC inference mismatch"
    fixed_dofs = np.array(sorted(fix.keys()), dtype=np.int64)
    np.savez_compressed(
        os.path.join(HERE, "embankment.npz"),
        nocoord=pts,
        elNodes=(cells + 1).astype(np.int64),
        fixed_dofs=fixed_dofs,
        fixed_vals=np.zeros(len(fixed_dofs)),
        material=np.array([E, nu, rho]),
        out_table=table,
        out_loadsum=np.array(head["loadsum"]),
        **{"vtk_" + re.sub(r"\W", "", k): v for k, v in data.items()},
    )
    json.dump({"name": name, "inp": inp, "source": "reference output files/*.vtk,*.out; control files/*.inp; freeCAD files/*.FCStd"},
              open(os.path.join(HERE, "embankment.json"), "w"), indent=1)
    print(name, "ne", len(cells), "nn", len(pts), "fixed dofs", len(fixed_dofs), "E nu rho", E, nu, rho)


def vtk_only(name, key):
    """Mesh + final fields of a reference result file (no boundary conditions): used for the
    checks that need none (principal stresses; stress recovery + nodal mapping of elastic runs;
    equilibrium K u = 0 at unloaded free nodes)."""
    pts, cells, data = read_vtk(os.path.join(REF, "output files", name + ".vtk"))
    E, nu, rho, _ = material_from_fcstd(os.path.join(REF, "freeCAD files", name + ".FCStd"))
    inp = [ln.strip() for ln in open(os.path.join(REF, "control files", name + ".inp"))]
    inp = [s for s in inp if s != ""][:21]
    head, table = parse_out(os.path.join(REF, "output files", name + ".out"))
    np.savez_compressed(
        os.path.join(HERE, key + ".npz"),
        nocoord=pts, elNodes=(cells + 1).astype(np.int64), material=np.array([E, nu, rho]),
        out_table=table, out_loadsum=np.array(head.get("loadsum", [])),
        **{"vtk_" + re.sub(r"\W", "", k): v for k, v in data.items()},
    )
    json.dump({"name": name, "inp": inp}, open(os.path.join(HERE, key + ".json"), "w"), indent=1)
    print(name, "ne", len(cells), "nn", len(pts), "E nu rho", E, nu, rho, "fields", list(data))


def lug_pressure():
    """Lug_Pressure_Example: the document holds geometry + constraints but no mesh, the .vtk holds the mesh the
    reference solved.  Constraint node sets are resolved on the vtk mesh from the document's BREP shapes
    (ConstraintFixed on Extrude:Face1, ConstraintPressure 100 N/mm2 on the two half-cylinders Fusion001:Face15/16 of
    the hole), the pressure faces are the 6-node tet faces lying on those cylinders, oriented with the outward
    normal like the surface elements FreeCAD hands to the reference (fcVM.py:270-287; sign -1 = not reversed).
    Cross-checks: the fixed nodes are exactly the nodes with zero displacement in the reference's result, and the
    load sum of the .out header is reproduced by the face integrals (tests/test_oracle_golden.py).  The .out of
    this model has 61 rows: two passes of nstep = 30 (continuation after the plot window, fcVM.py:1304-1308, 1615)."""
    from fcvm_hip import fcstd, meshgen
    from fcvm_hip.brep import BrepShape
    name = "Lug_Pressure_Example"
    vtk_only(name, "lug")
    g = dict(np.load(os.path.join(HERE, "lug.npz")))
    xyz, el = g["nocoord"], g["elNodes"]
    z = zipfile.ZipFile(os.path.join(REF, "freeCAD files", name + ".FCStd"))
    xml = z.read("Document.xml").decode("utf8", "replace")
    objs = fcstd._objects(xml)

    def shape_of(obj):
        m = re.search(r'<Part [^>]*file="([^"]+\.brp)"', fcstd._prop(objs[obj][1], "Shape") or "")
        return BrepShape(z.read(m.group(1)).decode("ascii", "replace"))

    refs = {n: re.findall(r'<Link obj="([^"]+)" sub="([^"]+)"', fcstd._prop(b, "References") or "") for n, (t, b) in objs.items()
            if t.startswith("Fem::Constraint")}
    assert refs == {"ConstraintFixed001": [("Extrude", "Face1")], "ConstraintPressure": [("Fusion001", "Face16"), ("Fusion001", "Face15")]}
    assert fcstd._float(objs["ConstraintPressure"][1], "Pressure") == 100.0 and not fcstd._bool(objs["ConstraintPressure"][1], "Reversed")
    tol = 1e-6 * float(np.ptp(xyz, axis=0).max())
    fixed = np.where(shape_of("Extrude").element("Face1").nodes(xyz, tol))[0]
    assert set(fixed) == set(np.where((g["vtk_Displacement"] == 0.0).all(1))[0])
    fus = shape_of("Fusion001")
    on = np.zeros(len(xyz), bool)
    for f in ("Face16", "Face15"):
        sub = fus.element(f)
        assert sub.surface[0] == "cylinder"
        on |= sub.nodes(xyz, tol)
    faces = []
    for f in meshgen.TET_FACES:
        opp = [k for k in range(4) if k not in f[:3]][0]
        for e in np.where(on[el[:, list(f)] - 1].all(1))[0]:
            n6 = el[e, list(f)]
            p, o = xyz[n6[:3] - 1], xyz[el[e, opp] - 1]
            if np.dot(np.cross(p[1] - p[0], p[2] - p[0]), p[0] - o) < 0:   # normal towards the element: turn the face round
                n6 = n6[[0, 2, 1, 5, 4, 3]]
            faces.append(n6)
    faces = np.array(faces, dtype=np.int64)
    g["fixed_dofs"] = np.array(sorted(3 * int(n) + k for n in fixed for k in range(3)), dtype=np.int64)
    g["pfaces"], g["pvals"] = faces, np.full(len(faces), -100.0)
    np.savez_compressed(os.path.join(HERE, "lug.npz"), **g)
    print(name, "fixed nodes", len(fixed), "pressure faces", len(faces), "rows", len(g["out_table"]))


def barbuckling():
    """GNLY (geometrically non-linear) reference run: 1000 x 10 x 10 bar (Part::Box), clamped at x = 0,
    axial force at x = L, imperfection of 10 mm along the first buckling shape.  Mesh, constraints and
    loads come from the FreeCAD document through fcvm_hip.fcstd (box faces are planes); the reference
    wrote the IMPERFECT coordinates into its .vtk (fcVM.py:1240, 2912), so the imperfection it drew
    from its ARPACK eigenvectors is recovered exactly as vtk points - mesh coordinates."""
    from fcvm_hip import fcstd
    name = "Bar buckling"
    pts, cells, data = read_vtk(os.path.join(REF, "output files", name + ".vtk"))
    head, table = parse_out(os.path.join(REF, "output files", name + ".out"))
    ev = [float(v) for v in re.search(r"elastic buckling factors:\s*\[([^\]]*)\]", open(os.path.join(REF, "output files", name + ".out")).read()).group(1).split()]
    inp = [ln.strip() for ln in open(os.path.join(REF, "control files", name + ".inp"))]
    inp = [s for s in inp if s != ""][:21]
    m = fcstd.load_model(os.path.join(REF, "freeCAD files", name + ".FCStd"))
    assert np.array_equal(m["elNodes"], cells + 1)
    fd = np.array(sorted(m["fix"]), dtype=np.int64)
    np.savez_compressed(
        os.path.join(HERE, "barbuckling.npz"),
        nocoord0=m["nocoord"], nocoord=pts, elNodes=m["elNodes"], fixed_dofs=fd,
        fixed_vals=np.array([m["fix"][int(d)] for d in fd]), ufaces=m["loads"].ufaces, uloads=m["loads"].uloads,
        material=m["material"], out_table=table, out_loadsum=np.array(head["loadsum"]), buckling_factors=np.array(ev),
        **{"vtk_" + re.sub(r"\W", "", k): v for k, v in data.items()},
    )
    json.dump({"name": name, "inp": inp}, open(os.path.join(HERE, "barbuckling.json"), "w"), indent=1)
    print(name, "ne", len(cells), "nn", len(pts), "buckling factors", ev, "max imperfection", np.abs(pts - m["nocoord"]).max())


def buckling():
    """Linear buckling reference run (GNLY, nstep = 1) of a 1000 x 10 x 10 cantilever with an end block.
    The mesh embedded in Buckling.FCStd is a later one (206 elements) than the one the shipped results
    were computed on (214), so mesh and fields come from the .vtk, the constraints from the document
    (ConstraintFixed on x = 0; ConstraintDisplacement yFree = false on the y = 0 faces of the four
    BooleanFragments solids; ConstraintForce 100 N along -x on one end face), resolved geometrically and
    cross-checked against the reference's own elastic displacement field: fixed dofs are exactly its
    exact zeros, and the loaded face is where K u has its only non-zero residual (checked in
    tests/test_oracle_golden.py to 1e-10: uniform traction -1 N/mm^2 on x = 1010, 0 <= z <= 10)."""
    from fcvm_hip import meshgen
    name = "Buckling"
    pts, cells, data = read_vtk(os.path.join(REF, "output files", name + ".vtk"))
    head, table = parse_out(os.path.join(REF, "output files", name + ".out"))
    ev = [float(v) for v in re.search(r"elastic buckling factors:\s*\[([^\]]*)\]", open(os.path.join(REF, "output files", name + ".out")).read()).group(1).split()]
    E, nu, rho, xml = material_from_fcstd(os.path.join(REF, "freeCAD files", name + ".FCStd"))
    inp = [ln.strip() for ln in open(os.path.join(REF, "control files", name + ".inp"))]
    inp = [s for s in inp if s != ""][:21]
    assert xml.count('<Object type="Fem::ConstraintFixed"') == 1 and xml.count('<Object type="Fem::ConstraintDisplacement"') == 1
    el = (cells + 1).astype(np.int64)
    fix = {}
    for n in np.where(np.abs(pts[:, 0]) < 1e-9)[0]:
        for k in range(3):
            fix[3 * int(n) + k] = 0.0
    for n in np.where(np.abs(pts[:, 1]) < 1e-9)[0]:
        fix[3 * int(n) + 1] = 0.0
    u = data["Elastic_Displacement"].reshape(-1)
    assert set(np.where(u == 0.0)[0].tolist()) == set(fix), "BC inference mismatch"
    faces = meshgen.boundary_faces_on_plane(el, pts, 0, pts[:, 0].max())
    zc = pts[faces[:, :3] - 1, 2].mean(1)
    faces = faces[(zc > 0.0) & (zc < 10.0)]
    area = 100.0
    assert abs(head["loadsum"][0] + 100.0) < 1e-9
    fd = np.array(sorted(fix), dtype=np.int64)
    np.savez_compressed(
        os.path.join(HERE, "buckling.npz"),
        nocoord=pts, elNodes=el, fixed_dofs=fd, fixed_vals=np.zeros(len(fd)), ufaces=faces,
        uloads=np.tile([head["loadsum"][0] / area, 0.0, 0.0], (len(faces), 1)), material=np.array([E, nu, rho]),
        out_table=table, out_loadsum=np.array(head["loadsum"]), buckling_factors=np.array(ev),
        **{"vtk_" + re.sub(r"\W", "", k): v for k, v in data.items()},
    )
    json.dump({"name": name, "inp": inp}, open(os.path.join(HERE, "buckling.json"), "w"), indent=1)
    print(name, "ne", len(cells), "nn", len(pts), "fixed dofs", len(fd), "loaded faces", len(faces), "factors", ev)


def u100():
    """GNLY run of an extruded U-profile column (gravity + axial and lateral end forces, imperfection 1 mm along the
    first buckling shape).  The document holds geometry (BREP), constraints and material but no mesh; the mesh is
    the one of the shipped .vtk, whose points carry the imperfection.  That file was written by the revision of the
    reference that scales the buckling shapes to max 1, so the imperfection (fcVM.py:1226-1236 with ev2 = 0) is
    sign * shape1 and the perfect coordinates are points - imperfection: they land on the CAD faces to 1e-13 and the
    clamped CAD face selects exactly the nodes whose displacement is exactly zero (both asserted here).  The .out
    table continues past the target load 0.8 after an interactive continuation; rows up to the target are kept."""
    from fcvm_hip import brep, meshgen
    name = "buckling_U100_3D"
    pts, cells, data = read_vtk(os.path.join(REF, "output files", name + ".vtk"))
    head, table = parse_out(os.path.join(REF, "output files", name + ".out"))
    ev = [float(v) for v in re.search(r"elastic buckling factors:\s*\[([^\]]*)\]", open(os.path.join(REF, "output files", name + ".out")).read()).group(1).split()]
    E, nu, rho, xml = material_from_fcstd(os.path.join(REF, "freeCAD files", name + ".FCStd"))
    inp = [ln.strip() for ln in open(os.path.join(REF, "control files", name + ".inp"))]
    inp = [s for s in inp if s != ""][:21]
    b = brep.BrepShape(zipfile.ZipFile(os.path.join(REF, "freeCAD files", name + ".FCStd")).read("PartShape2.brp").decode())
    s1 = data["Buckling_shape_for_lambda1_=_3.895"].reshape(-1)
    assert abs(np.abs(s1).max() - 1.0) < 1e-12 and float(inp[18]) == 1.0 and float(inp[20]) == 0.0
    imper = np.sign(s1[np.argmax(np.abs(s1))]) * s1
    x0 = pts - imper.reshape(-1, 3)
    el = (cells + 1).astype(np.int64)
    bfaces = np.sort(np.concatenate([el[:, list(f)] for f in meshgen.TET_FACES]), axis=1)
    u, c = np.unique(bfaces, axis=0, return_counts=True)
    bn = np.unique(u[c == 1]) - 1
    on_cad = np.zeros(len(x0), bool)
    for n in range(1, len(b.faces) + 1):
        on_cad |= b.face(n).nodes(x0, 1e-6)
    assert on_cad[bn].all() and on_cad.sum() == len(bn), "recovered geometry does not sit on the CAD faces"
    # constraints of the document: ConstraintFixed on Face11, two ConstraintForce on Face10 (50 kN along -z, 2 kN along +x)
    assert re.findall(r'<Link obj="Extrude" sub="(Face\d+)"', xml) == ["Face10", "Face10", "Face11"]
    fixed = np.where(b.face(11).nodes(x0, 1e-6))[0]
    assert set(fixed) == set(np.where((data["Elastic_Displacement"] == 0.0).all(1))[0])
    fd = np.array(sorted(3 * int(n) + k for n in fixed for k in range(3)), dtype=np.int64)
    on = b.face(10).nodes(x0, 1e-6)
    tf = np.concatenate([el[:, list(f)] for f in meshgen.TET_FACES])
    tf = tf[on[tf - 1].all(1)]
    area = b.face(10).area
    nrow = int(np.argmax(np.isclose(table[:, 4], float(inp[14])))) + 1   # rows up to the target load factor
    # the selection of the shipped .avr summary (fcVM_sum.FCMacro: Edge7 and Face2 of the Extrude): the 3-node line elements /
    # 6-node triangles of the mesh on those CAD entities, as FemMesh.getEdgesByEdge / getFacesByFace hand them to calcSum
    avr = open(os.path.join(REF, "output files", name + ".avr")).read().splitlines()
    assert avr[6].split()[-1] == "Extrude(Extrude).Edge7" and avr[11].split()[-1] == "Extrude(Extrude).Face2"
    on_e, on_f = b.edge(7).nodes(x0, 1e-6), b.face(2).nodes(x0, 1e-6)
    te = np.concatenate([el[:, [p_, q_, 4 + k]] for k, (p_, q_) in enumerate(meshgen.MID_PAIRS)])
    te = te[on_e[te - 1].all(1)]
    te = te[np.unique(np.sort(te[:, :2], axis=1), axis=0, return_index=True)[1]]
    tf2 = np.concatenate([el[:, list(f)] for f in meshgen.TET_FACES])
    tf2 = tf2[on_f[tf2 - 1].all(1)]
    np.savez_compressed(
        os.path.join(HERE, "u100.npz"),
        nocoord0=x0, nocoord=pts, elNodes=el, fixed_dofs=fd, fixed_vals=np.zeros(len(fd)), ufaces=tf,
        uloads=np.tile([2000.0 / area, 0.0, -50000.0 / area], (len(tf), 1)), material=np.array([E, nu, rho]),
        out_table=table[:nrow], out_table_full=table, out_loadsum=np.array(head["loadsum"]), buckling_factors=np.array(ev),
        shape1=data["Buckling_shape_for_lambda1_=_3.895"], shape2=data["Buckling_shape_for_lambda2_=_8.394"],
        elastic_displacement=data["Elastic_Displacement"],
        avr_edge7=te, avr_face2=tf2, vtk_csr=data["Critical_Strain_Ratio\n"], vtk_peeq=data["Equivalent_Plastic_Strain\n"],
        vtk_svm=data["von_Mises_Stress\n"])
    json.dump({"name": name, "inp": inp, "avr": avr}, open(os.path.join(HERE, "u100.json"), "w"), indent=1)
    print(name, "ne", len(cells), "nn", len(pts), "factors", ev, "rows kept", nrow, "loaded faces", len(tf))


def out_only(name, key):
    """Control file + .out table of a model whose document holds no mesh (BASELINE configs[1])."""
    head, table = parse_out(os.path.join(REF, "output files", name + ".out"))
    E, nu, rho, _ = material_from_fcstd(os.path.join(REF, "freeCAD files", name + ".FCStd"))
    inp = [ln.strip() for ln in open(os.path.join(REF, "control files", name + ".inp"))]
    json.dump({"name": name, "inp": [s for s in inp if s != ""][:21], "ne": head["ne"], "nn": head["nn"], "loadsum": head["loadsum"],
               "material": [E, nu, rho], "out_table": table.tolist()}, open(os.path.join(HERE, key + ".json"), "w"), indent=1)
    print(name, "ne", head["ne"], "rows", len(table))


def fcstd_fixture(name, key, members=("Document.xml", "FemMesh.unv")):
    """Document.xml + FemMesh.unv (+ the BREP geometry file of the meshed shape) of a reference FreeCAD document
    (data: model definition + mesh), re-zipped without thumbnails / result arrays."""
    src = zipfile.ZipFile(os.path.join(REF, "freeCAD files", name + ".FCStd"))
    with zipfile.ZipFile(os.path.join(HERE, key + ".FCStd"), "w", zipfile.ZIP_DEFLATED) as out:
        for member in members:
            out.writestr(zipfile.ZipInfo(member, date_time=(2025, 1, 1, 0, 0, 0)), src.read(member), zipfile.ZIP_DEFLATED)
    print(name, "->", key + ".FCStd")


if __name__ == "__main__":
    fcstd_fixture("tensile", "tensile")
    fcstd_fixture("Embankment_with_Ditch_Example", "embankment", ("Document.xml", "FemMesh.unv", "Extrude.Shape.brp"))
    shutil.copyfile(os.path.join(REF, "output files", "Embankment_with_Ditch_Example.out"), os.path.join(HERE, "embankment_reference.out"))
    fcstd_fixture("Plate_with_hole_Example", "plate_with_hole", ("Document.xml", "PartShape2.brp"))  # no mesh in this document
    # further documents with an embedded mesh (no shipped result files): inputs for GPU-vs-oracle runs of the whole chain
    sweep = {}
    for name, key, members in (("Lug_FaceLoad_Example", "lug_faceload", None), ("Simple Shear", "simple_shear", None),
                               ("Block_Disp_Control_Example", "block_disp", None), ("Elastica", "elastica", None)):
        src = zipfile.ZipFile(os.path.join(REF, "freeCAD files", name + ".FCStd"))
        xml = src.read("Document.xml").decode("utf8", "replace")
        mesh = [m for n_, m in re.findall(r'<Object name="(FEMMesh[^"]*)".*?<FemMesh file="([^"]+)"', xml[xml.find("<ObjectData"):], re.S)][-1]
        brps = sorted(set(re.findall(r'file="([^"]+\.brp)"', xml)) & set(src.namelist()))
        fcstd_fixture(name, key, ("Document.xml", mesh) + tuple(b for b in brps if len(src.read(b)) > 0))
        inp = [ln.strip() for ln in open(os.path.join(REF, "control files", name + ".inp"))]
        sweep[key] = {"name": name, "inp": [s for s in inp if s != ""][:21]}
    json.dump(sweep, open(os.path.join(HERE, "documents.json"), "w"), indent=1)
    fcstd_fixture("VM_Uniaxial_Tension_Example", "vm_uniaxial")  # BASELINE configs[0]
    inp = [ln.strip() for ln in open(os.path.join(REF, "control files", "VM_Uniaxial_Tension_Example.inp"))]
    json.dump({"name": "VM_Uniaxial_Tension_Example", "inp": [s for s in inp if s != ""]}, open(os.path.join(HERE, "vm_uniaxial.json"), "w"), indent=1)
    barbuckling()
    embankment()
    vtk_only("tensile", "tensile")
    buckling()
    u100()
    out_only("Plate_with_hole_Example", "plate_with_hole")
    lug_pressure()
