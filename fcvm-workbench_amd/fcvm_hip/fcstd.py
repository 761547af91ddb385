"""Reading a FreeCAD document (.FCStd) without FreeCAD: the mesh (FemMesh.unv inside the zip) and
the FEM constraints of Document.xml, for models bounded by planes and cylinders.

This covers what the reference's setUpInput does through the FreeCAD API
(reference: source code/fcVM.py:122-347): element / node arrays in fcVM's internal node order,
the `fix` dictionary, load faces / edges / vertices with their tractions and the material triple.  Sub-shape
references (FaceN / EdgeN / VertexN) are resolved from the shape's BREP file stored in the document
(fcvm_hip.brep); when that file is missing, face references of a
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


class _BoxShape:
    """Sub-shapes of an untransformed-axes Part::Box from its Length / Width / Height / Placement (documents whose
    PartShape.brp is not at hand): FreeCAD / OpenCASCADE numbering, see the module docstring."""

    def __init__(self, body):
        self.dims = np.array([_float(body, "Length"), _float(body, "Width"), _float(body, "Height")])
        pl = _prop(body, "Placement")
        self.org = np.array([float(re.search(r'P%s="([^"]+)"' % a, pl).group(1)) for a in "xyz"])
        if abs(float(re.search(r'A="([^"]+)"', pl).group(1))) > 1e-12:
            raise NotImplementedError("rotated boxes need the document's BREP file")

    def element(self, name):
        from .brep import SubShape
        m = re.match(r"^(Face|Vertex)(\d+)$", name)
        if not m:
            raise NotImplementedError("reference %s: only Face1..Face6 / Vertex1..Vertex8 of a box without BREP file" % name)
        k = int(m.group(2)) - 1
        sub = SubShape(m.group(1).lower(), name)
        if m.group(1) == "Vertex":
            if not 0 <= k < 8:
                raise NotImplementedError("reference %s: a box has Vertex1..Vertex8" % name)
            sub.point = self.org + self.dims * np.array([(k >> 2) & 1, (k >> 1) & 1, 1 - (k & 1)])
            return sub
        if not 0 <= k < 6:
            raise NotImplementedError("reference %s: a box has Face1..Face6" % name)
        ax = k // 2
        lo, hi = self.org.copy(), self.org + self.dims
        lo[ax] = hi[ax] = self.org[ax] + (self.dims[ax] if k % 2 else 0.0)
        b, c = [i for i in range(3) if i != ax]
        corners = []
        for sb, sc in ((0, 0), (1, 0), (1, 1), (0, 1)):
            q = lo.copy()
            q[b] = hi[b] if sb else lo[b]
            q[c] = hi[c] if sc else lo[c]
            corners.append(q)
        corners = np.array(corners)
        nrm = np.zeros(3)
        nrm[ax] = 1.0
        sub.surface = ("plane", lo, nrm)
        sub.segments = np.stack([corners, np.roll(corners, -1, 0)], 1)
        return sub


def load_model(path_or_bytes, mesh_member=None):
    """Returns dict(name, elNodes, nocoord, material [E, nu, rho], fix, loads (Loads)).

    What the reference's setUpInput extracts through the FreeCAD API (fcVM.py:122-347): the 2nd-order mesh of the
    FEMMesh* object, node sets / face / edge elements of the sub-shapes the constraints reference -- resolved from
    the shape's BREP file in the document (fcvm_hip.brep: planes, cylinders, straight and circular edges), or
    analytically for a Part::Box whose BREP file is missing."""
    from .brep import BrepShape
    zf = zipfile.ZipFile(io.BytesIO(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else path_or_bytes)
    members = set(zf.namelist())
    xml = zf.read("Document.xml").decode("utf8", "replace")
    objs = _objects(xml)
    if mesh_member is None:
        mesh_member = "FemMesh.unv"
        for name, (typ, body) in objs.items():
            m = re.search(r'<FemMesh file="([^"]+)"', body)
            if name.startswith("FEMMesh") and m:   # fcVM.py:97-99: the object the reference picks
                mesh_member = m.group(1)
    labels, xyz, els = read_unv(zf.read(mesh_member).decode("ascii", "replace"))
    if 118 not in els:
        raise ValueError("no 10-node tetrahedra in %s (fcVM needs 2nd-order volume meshes)" % mesh_member)
    lab2idx = np.full(labels.max() + 1, -1, dtype=np.int64)
    lab2idx[labels] = np.arange(len(labels)) + 1  # 1-based node numbers in file order (FemMesh.Nodes order)
    elNodes = lab2idx[els[118]][:, TET10_PERM]
    tri = lab2idx[els[42]][:, TRIA6_PERM] if 42 in els else np.zeros((0, 6), np.int64)
    seg = lab2idx[els[22]][:, SEG3_PERM] if 22 in els else np.zeros((0, 3), np.int64)
    tol = 1e-6 * float(np.ptp(xyz, axis=0).max())
    shapes = {}

    def shape_of(obj):
        if obj not in shapes:
            typ, body = objs[obj]
            m = re.search(r'<Part [^>]*file="([^"]+\.brp)"', _prop(body, "Shape") or "")
            if m and m.group(1) in members and len(zf.read(m.group(1))) > 0:
                shapes[obj] = BrepShape(zf.read(m.group(1)).decode("ascii", "replace"))
            elif typ == "Part::Box":
                shapes[obj] = _BoxShape(body)
            else:
                raise NotImplementedError("object %s (%s): its BREP file is not in the document" % (obj, typ))
        return shapes[obj]

    def refs(body):
        out = []
        for obj, sub in re.findall(r'<Link obj="([^"]+)" sub="([^"]+)"', _prop(body, "References") or ""):
            out.append(shape_of(obj).element(sub))
        return out

    def nodes_of(sub):
        return np.where(sub.nodes(xyz, tol))[0]

    fix, pf, pv, uf, ul, ed, el_, vn, vl = {}, [], [], [], [], [], [], [], []
    for name, (typ, body) in objs.items():
        if typ in ("Fem::ConstraintFixed", "Fem::ConstraintDisplacement"):  # fcVM.py:187-254
            free = [False] * 3 if typ.endswith("Fixed") else [_bool(body, a + "Free", True) for a in "xyz"]
            val = [0.0] * 3 if typ.endswith("Fixed") else [_float(body, a + "Displacement", 0.0) for a in "xyz"]
            for sub in refs(body):
                for n in nodes_of(sub):
                    for k in range(3):
                        if not free[k]:
                            fix[3 * int(n) + k] = val[k]
        elif typ == "Fem::ConstraintForce":  # fcVM.py:289-324: force spread over the referenced vertices / length / area
            F = _float(body, "Force")
            dv = _prop(body, "DirectionVector")
            d = np.array([float(re.search(r'value%s="([^"]+)"' % a, dv).group(1)) for a in "XYZ"])
            subs = refs(body)
            N = sum(1 for s_ in subs if s_.kind == "vertex")
            L = sum(s_.length for s_ in subs if s_.kind == "edge")
            A = sum(s_.area for s_ in subs if s_.kind == "face")
            for sub in subs:
                on = sub.nodes(xyz, tol)
                if sub.kind == "vertex":
                    idx = np.where(on)[0]
                    vn.append(idx + 1)
                    vl.append(np.tile(F * d / N, (len(idx), 1)))
                elif sub.kind == "edge":
                    es = seg[on[seg - 1].all(1)]
                    ed.append(es)
                    el_.append(np.tile(F * d / L, (len(es), 1)))
                else:
                    fs = tri[on[tri - 1].all(1)]
                    uf.append(fs)
                    ul.append(np.tile(F * d / A, (len(fs), 1)))
        elif typ == "Fem::ConstraintPressure":  # fcVM.py:270-287
            sign = 1.0 if _bool(body, "Reversed") else -1.0
            p = _float(body, "Pressure")
            for sub in refs(body):
                if sub.kind != "face":
                    continue
                on = sub.nodes(xyz, tol)
                fs = tri[on[tri - 1].all(1)]
                pf.append(fs)
                pv.append(np.full(len(fs), sign * p))
    mat = _prop([b for _, (t, b) in objs.items() if t == "App::MaterialObjectPython"][0], "Material")
    item = lambda k: re.search(r'<Item key="%s" value="([^"]*)"' % k, mat).group(1)  # noqa: E731
    E = _quantity(item("YoungsModulus"), {"MPa": 1.0, "kPa": 1e-3, "GPa": 1e3, "Pa": 1e-6, "kg/(mm*s^2)": 1e-3, "N/mm^2": 1.0, "N/m^2": 1e-6})
    rho = _quantity(item("Density"), {"kg/m^3": 1e-9, "kg/mm^3": 1.0})
    cat = lambda a, shape: np.concatenate(a, axis=0) if a else np.zeros(shape)  # noqa: E731
    loads = Loads(pfaces=cat(pf, (0, 6)), pvals=cat(pv, (0,)), ufaces=cat(uf, (0, 6)), uloads=cat(ul, (0, 3)),
                  edges=cat(ed, (0, 3)), eloads=cat(el_, (0, 3)), vnodes=cat(vn, (0,)), vloads=cat(vl, (0, 3)))
    label = re.search(r'<Property name="Label"[^>]*>\s*<String value="([^"]*)"', xml)
    return dict(name=label.group(1) if label else "model", elNodes=elNodes.astype(np.int64), nocoord=xyz,
                material=np.array([E, float(item("PoissonRatio")), rho]), fix=fix, loads=loads)
