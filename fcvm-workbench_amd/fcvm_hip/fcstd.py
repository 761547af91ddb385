"""Reading a FreeCAD document (.FCStd) without FreeCAD: the mesh (FemMesh.unv inside the zip) and
the FEM constraints of Document.xml, for models whose geometry is a Part::Box.

This covers, for box-shaped models, what the reference's setUpInput does through the FreeCAD API
(reference: source code/fcVM.py:122-347): element / node arrays in fcVM's internal node order,
the `fix` dictionary, load faces with their tractions and the material triple.  Face references of a
Part::Box are planes (OpenCASCADE numbering: Face1 x=0, Face2 x=L, Face3 y=0, Face4 y=W, Face5 z=0,
Face6 z=H, then translated by the Placement) and its corners are Vertex1..8 = (0,0,H) (0,0,0) (0,W,H)
(0,W,0) (L,0,H) (L,0,0) (L,W,H) (L,W,0) (the order of the vertex records in the document's
PartShape.brp); other shapes need FreeCAD's geometry kernel and raise NotImplementedError.

UNV -> fcVM node order (derived from the reference's own data: the FemMesh.unv stored in
tensile.FCStd against the cells of tensile.vtk, identical for all elements):
  tet10  (fe 118): UNV (c0,m01,c1,m12,c2,m02,m03,m13,m23,c3) -> (c0,c1,c2,c3,m01,m12,m02,m03,m13,m23)
  tria6  (fe 42):  UNV (c0,m01,c1,m12,c2,m02)                -> (c0,c1,c2,m01,m12,m02)
  seg3   (fe 22):  UNV (e0,mid,e1)                            -> (e0,e1,mid)
"""
import io
import re
import zipfile

import numpy as np

from .loads import Loads

TET10_PERM = (0, 2, 4, 9, 1, 3, 5, 6, 7, 8)
TRIA6_PERM = (0, 2, 4, 1, 3, 5)
SEG3_PERM = (0, 2, 1)
_BEAMS = (11, 21, 22, 23, 24)


def read_unv(text):
    """Universal-file datasets 2411 (nodes) and 2412 (elements).  Returns (labels [n], xyz [n,3],
    {fe_descriptor: int array [m, nnodes]} with node LABELS)."""
    lines = text.splitlines()
    i, labels, xyz, els = 0, [], [], {}
    while i < len(lines):
        if lines[i].strip() == "-1" and i + 1 < len(lines):
            ds = lines[i + 1].strip()
            i += 2
            if ds == "2411":
                while lines[i].strip() != "-1":
                    labels.append(int(lines[i].split()[0]))
                    xyz.append([float(v.replace("D", "E")) for v in lines[i + 1].split()])
                    i += 2
            elif ds == "2412":
                while lines[i].strip() != "-1":
                    t = lines[i].split()
                    fe, nn = int(t[1]), int(t[5])
                    i += 1
                    if fe in _BEAMS:
                        i += 1
                    nd = []
                    while len(nd) < nn:
                        nd += [int(v) for v in lines[i].split()]
                        i += 1
                    els.setdefault(fe, []).append(nd)
            else:
                while i < len(lines) and lines[i].strip() != "-1":
                    i += 1
            i += 1
        else:
            i += 1
    return np.array(labels), np.array(xyz, dtype=np.float64), {k: np.array(v, dtype=np.int64) for k, v in els.items()}


def _objects(xml):
    body = xml[xml.find("<ObjectData"):]
    types = {name: typ for typ, name in re.findall(r'<Object type="([^"]+)" name="([^"]+)"', xml)}
    out = {}
    for name in types:
        m = re.search(r'<Object name="%s"[^>]*>(.*?)</Object>' % re.escape(name), body, re.S)
        out[name] = (types[name], m.group(1) if m else "")
    return out


def _prop(body, name):
    m = re.search(r'<Property name="%s"[^>]*>(.*?)</Property>' % name, body, re.S)
    return m.group(1) if m else None


def _float(body, name, default=None):
    p = _prop(body, name)
    if p is None:
        return default
    m = re.search(r'value="([^"]+)"', p)
    return float(m.group(1)) if m else default


def _bool(body, name, default=False):
    p = _prop(body, name)
    if p is None:
        return default
    return 'value="true"' in p


def _quantity(s, units):
    v, u = s.split()
    return float(v) * units[u]


def load_model(path_or_bytes, mesh_member="FemMesh.unv"):
    """Returns dict(name, elNodes, nocoord, material [E, nu, rho], fix, loads (Loads), box)."""
    zf = zipfile.ZipFile(io.BytesIO(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else path_or_bytes)
    xml = zf.read("Document.xml").decode("utf8", "replace")
    objs = _objects(xml)
    boxes = {n: b for n, (t, b) in objs.items() if t == "Part::Box"}
    if len(boxes) != 1 or any(t.startswith("Part::") and t != "Part::Box" for t, _ in objs.values()):
        raise NotImplementedError("only single Part::Box geometries can be read without FreeCAD")
    (bname, bbody), = boxes.items()
    dims = np.array([_float(bbody, "Length"), _float(bbody, "Width"), _float(bbody, "Height")])
    pl = _prop(bbody, "Placement")
    org = np.array([float(re.search(r'P%s="([^"]+)"' % a, pl).group(1)) for a in "xyz"])
    if abs(float(re.search(r'A="([^"]+)"', pl).group(1))) > 1e-12:
        raise NotImplementedError("rotated boxes are not supported")
    labels, xyz, els = read_unv(zf.read(mesh_member).decode("ascii", "replace"))
    if 118 not in els:
        raise ValueError("no 10-node tetrahedra in %s (fcVM needs 2nd-order volume meshes)" % mesh_member)
    lab2idx = np.full(labels.max() + 1, -1, dtype=np.int64)
    lab2idx[labels] = np.arange(len(labels)) + 1  # 1-based node numbers in file order (FemMesh.Nodes order)
    elNodes = lab2idx[els[118]][:, TET10_PERM]
    tri = lab2idx[els[42]][:, TRIA6_PERM] if 42 in els else np.zeros((0, 6), np.int64)
    tol = 1e-6 * dims.max()

    def plane(face):
        if not face.startswith("Face") or not face[4:].isdigit() or not 0 <= int(face[4:]) - 1 < 6:
            raise NotImplementedError("reference %s: only Face1..Face6 / Vertex1..Vertex8 of the box are supported" % face)
        k = int(face[4:]) - 1
        ax = k // 2
        return ax, org[ax] + (dims[ax] if k % 2 else 0.0)

    def nodes_of(face):
        if face.startswith("Vertex"):
            k = int(face[6:]) - 1
            if not 0 <= k < 8:
                raise NotImplementedError("reference %s: a box has Vertex1..Vertex8" % face)
            p = org + dims * np.array([(k >> 2) & 1, (k >> 1) & 1, 1 - (k & 1)])
            hit = np.where(np.abs(xyz - p).max(1) < tol)[0]
            if len(hit) != 1:
                raise ValueError("no mesh node at %s of the box" % face)
            return hit
        ax, v = plane(face)
        return np.where(np.abs(xyz[:, ax] - v) < tol)[0]

    def faces_of(face):
        ax, v = plane(face)
        on = np.abs(xyz[:, ax] - v) < tol
        return tri[on[tri[:, :3] - 1].all(1)]

    fix, pf, pv, uf, ul = {}, [], [], [], []
    for name, (typ, body) in objs.items():
        refs = re.findall(r'<Link obj="%s" sub="([^"]+)"' % re.escape(bname), _prop(body, "References") or "")
        if typ in ("Fem::ConstraintFixed", "Fem::ConstraintDisplacement"):  # fcVM.py:187-254
            free = [False] * 3 if typ.endswith("Fixed") else [_bool(body, a + "Free", True) for a in "xyz"]
            val = [0.0] * 3 if typ.endswith("Fixed") else [_float(body, a + "Displacement", 0.0) for a in "xyz"]
            for face in refs:
                for n in nodes_of(face):
                    for k in range(3):
                        if not free[k]:
                            fix[3 * int(n) + k] = val[k]
        elif typ == "Fem::ConstraintForce":  # fcVM.py:289-324: force spread over the referenced area
            F = _float(body, "Force")
            dv = _prop(body, "DirectionVector")
            d = np.array([float(re.search(r'value%s="([^"]+)"' % a, dv).group(1)) for a in "XYZ"])
            area = 0.0
            for face in refs:
                ax, _ = plane(face)
                area += np.prod(np.delete(dims, ax))
            for face in refs:
                fs = faces_of(face)
                uf.append(fs)
                ul.append(np.tile(F * d / area, (len(fs), 1)))
        elif typ == "Fem::ConstraintPressure":  # fcVM.py:270-287
            sign = 1.0 if _bool(body, "Reversed") else -1.0
            p = _float(body, "Pressure")
            for face in refs:
                fs = faces_of(face)
                pf.append(fs)
                pv.append(np.full(len(fs), sign * p))
    mat = _prop([b for _, (t, b) in objs.items() if t == "App::MaterialObjectPython"][0], "Material")
    item = lambda k: re.search(r'<Item key="%s" value="([^"]*)"' % k, mat).group(1)  # noqa: E731
    E = _quantity(item("YoungsModulus"), {"MPa": 1.0, "kPa": 1e-3, "GPa": 1e3, "Pa": 1e-6})
    rho = _quantity(item("Density"), {"kg/m^3": 1e-9, "kg/mm^3": 1.0})
    cat = lambda a, shape: np.concatenate(a, axis=0) if a else np.zeros(shape)  # noqa: E731
    loads = Loads(pfaces=cat(pf, (0, 6)), pvals=cat(pv, (0,)), ufaces=cat(uf, (0, 6)), uloads=cat(ul, (0, 3)))
    label = re.search(r'<Property name="Label"[^>]*>\s*<String value="([^"]*)"', xml)
    return dict(name=label.group(1) if label else "model", elNodes=elNodes.astype(np.int64), nocoord=xyz,
                material=np.array([E, float(item("PoissonRatio")), rho]), fix=fix, loads=loads,
                box=dict(origin=org, size=dims))
