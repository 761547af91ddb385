"""Reader for the OpenCASCADE ASCII BREP files inside a FreeCAD document (`<Part file="X.brp"/>`), as far as
the reference's `setUpInput` needs geometry (fcVM.py:199-215, 270-324): which mesh nodes lie on the referenced
`FaceN` / `EdgeN` / `VertexN` of a shape, and the area / length of those sub-shapes.

FreeCAD numbers sub-shapes by first encounter in a depth-first walk over the shape tree
(TopExp::MapShapes), two occurrences being the same sub-shape when TShape and location agree.  The same walk is
done here over the TShapes table (a TShape used twice with different locations -- e.g. the two caps of an
extrusion -- yields two faces).  Supported geometry: planes and cylinders, straight and circular edges;
anything else raises NotImplementedError when it is asked for.
"""
import re

import numpy as np

_FLAGS = re.compile(r"^[01]{7}$")


def _inv(T):
    R, t = T[:3, :3], T[:3, 3]
    Ti = np.eye(4)
    Ti[:3, :3] = R.T
    Ti[:3, 3] = -R.T @ t
    return Ti


class _Cursor:
    def __init__(self, tokens):
        self.t, self.i = tokens, 0

    def next(self):
        self.i += 1
        return self.t[self.i - 1]

    def floats(self, n):
        out = [float(v) for v in self.t[self.i:self.i + n]]
        self.i += n
        return out


def _read_curve(cur):
    """One 3D curve of the Curves table -> ('line', origin, dir) | ('circle', centre, axis, xdir, ydir, r) | None."""
    kind = int(cur.next())
    if kind == 1:
        v = cur.floats(6)
        return ("line", np.array(v[:3]), np.array(v[3:]))
    if kind == 2:
        v = cur.floats(13)
        return ("circle", np.array(v[:3]), np.array(v[3:6]), np.array(v[6:9]), np.array(v[9:12]), v[12])
    if kind in (3, 5):      # ellipse, hyperbola: axes + two radii
        cur.floats(14)
        return None
    if kind == 4:           # parabola
        cur.floats(13)
        return None
    if kind == 8:           # trimmed curve: first, last, basis curve
        cur.floats(2)
        return _read_curve(cur)
    raise NotImplementedError("BREP curve type %d (Bezier / B-spline / offset) is not supported" % kind)


def _read_surface(cur):
    """-> ('plane', origin, normal) | ('cylinder', origin, axis, r) | None (recognised, but not usable here)."""
    kind = int(cur.next())
    if kind == 1:
        v = cur.floats(12)
        return ("plane", np.array(v[:3]), np.array(v[3:6]))
    if kind == 2:
        v = cur.floats(13)
        return ("cylinder", np.array(v[:3]), np.array(v[3:6]), v[12])
    if kind in (3, 5):      # cone (radius, semi-angle), torus (two radii)
        cur.floats(14)
        return None
    if kind == 4:           # sphere
        cur.floats(13)
        return None
    if kind == 6:           # linear extrusion: direction + basis curve
        cur.floats(3)
        _read_curve(cur)
        return None
    if kind == 7:           # revolution: axis point + direction + basis curve
        cur.floats(6)
        _read_curve(cur)
        return None
    if kind == 10:          # rectangular trimmed: 4 bounds + basis surface
        cur.floats(4)
        return _read_surface(cur)
    raise NotImplementedError("BREP surface type %d (Bezier / B-spline / offset) is not supported" % kind)


class SubShape:
    """A face, edge or vertex in world coordinates."""

    def __init__(self, kind, name):
        self.kind, self.name = kind, name
        self.surface = None      # faces: ('plane', p, n) / ('cylinder', p, axis, r) / None
        self.segments = None     # faces / edges: boundary resp. the edge itself as straight pieces [m,2,3]
        self.point = None        # vertices

    # -- measures ------------------------------------------------------------------------------
    @property
    def length(self):
        s = self.segments
        return float(np.linalg.norm(s[:, 1] - s[:, 0], axis=1).sum())

    @property
    def area(self):
        if self.surface is None:
            raise NotImplementedError("%s: area of this surface type" % self.name)
        if self.surface[0] == "plane":
            _, p, n = self.surface
            a, b = self.segments[:, 0] - p, self.segments[:, 1] - p
            return float(abs(0.5 * np.einsum("ij,j->", np.cross(a, b), n)))  # oriented loops: outer minus holes
        _, p, ax, r = self.surface
        th, h = self._cyl_param(self.segments.reshape(-1, 3))
        lo, span = self._arc(th)
        return float(r * span * (h.max() - h.min()))

    # -- node selection ------------------------------------------------------------------------
    def nodes(self, xyz, tol):
        """Boolean mask of the points of xyz [n,3] that lie on this sub-shape."""
        if self.kind == "vertex":
            return np.abs(xyz - self.point).max(1) < tol
        if self.kind == "edge":
            return self._seg_dist(xyz) < tol
        if self.surface is None:
            raise NotImplementedError("%s lies on a surface type that is not supported" % self.name)
        if self.surface[0] == "plane":
            _, p, n = self.surface
            cand = np.abs((xyz - p) @ n) < tol
            idx = np.where(cand)[0]
            if len(idx) == 0:
                return cand
            e1 = np.cross(n, [1.0, 0.0, 0.0]) if abs(n[0]) < 0.9 else np.cross(n, [0.0, 1.0, 0.0])
            e1 /= np.linalg.norm(e1)
            e2 = np.cross(n, e1)
            q = np.stack([(xyz[idx] - p) @ e1, (xyz[idx] - p) @ e2], 1)
            a = np.stack([(self.segments[:, 0] - p) @ e1, (self.segments[:, 0] - p) @ e2], 1)
            b = np.stack([(self.segments[:, 1] - p) @ e1, (self.segments[:, 1] - p) @ e2], 1)
            # crossing number of the ray towards +x (all loops together: holes cancel), plus points on the boundary
            ya, yb = a[None, :, 1], b[None, :, 1]
            qy, qx = q[:, None, 1], q[:, None, 0]
            straddle = (ya > qy) != (yb > qy)
            with np.errstate(divide="ignore", invalid="ignore"):
                xi = a[None, :, 0] + (qy - ya) * (b[None, :, 0] - a[None, :, 0]) / (yb - ya)
            inside = (np.sum(straddle & (xi > qx), axis=1) % 2) == 1
            on_edge = self._seg_dist(xyz[idx]) < tol
            out = np.zeros(len(xyz), bool)
            out[idx] = inside | on_edge
            return out
        _, p, ax, r = self.surface
        th, h = self._cyl_param(xyz)
        d = xyz - p
        rad = np.linalg.norm(d - np.outer(d @ ax, ax), axis=1)
        bth, bh = self._cyl_param(self.segments.reshape(-1, 3))
        lo, span = self._arc(bth)
        dth = np.mod(th - lo, 2.0 * np.pi)
        on_arc = (dth <= span + tol / r) | (dth >= 2.0 * np.pi - tol / r)
        return (np.abs(rad - r) < tol) & on_arc & (h > bh.min() - tol) & (h < bh.max() + tol)

    def _seg_dist(self, x):
        a, b = self.segments[:, 0], self.segments[:, 1]
        ab = b - a
        den = np.maximum(np.einsum("ij,ij->i", ab, ab), 1e-300)
        t = np.clip(np.einsum("nij,ij->ni", x[:, None, :] - a[None], ab) / den, 0.0, 1.0)
        c = a[None] + t[:, :, None] * ab[None]
        return np.linalg.norm(x[:, None, :] - c, axis=2).min(1)

    def _cyl_param(self, x):
        _, p, ax, r = self.surface
        e1 = np.cross(ax, [1.0, 0.0, 0.0]) if abs(ax[0]) < 0.9 else np.cross(ax, [0.0, 1.0, 0.0])
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(ax, e1)
        d = x - p
        return np.arctan2(d @ e2, d @ e1), d @ ax

    @staticmethod
    def _arc(th):
        """smallest arc [lo, lo + span] containing all angles th"""
        s = np.sort(np.mod(th, 2.0 * np.pi))
        gaps = np.diff(np.concatenate([s, [s[0] + 2.0 * np.pi]]))
        k = int(np.argmax(gaps))
        if gaps[k] < 1e-6 + 2.0 * np.pi / max(len(s), 1) * 1.5:   # samples all around: the whole circumference
            return 0.0, 2.0 * np.pi
        return float(s[(k + 1) % len(s)]), float(2.0 * np.pi - gaps[k])


class BrepShape:
    def __init__(self, text, arc_points=180):
        self.arc_points = arc_points
        lines = text.splitlines()
        head = {}
        for i, ln in enumerate(lines):
            m = re.match(r"^(Locations|Curve2ds|Curves|Polygon3D|PolygonOnTriangulations|Surfaces|Triangulations|TShapes)\s+(\d+)\s*$", ln)
            if m and m.group(1) not in head:
                head[m.group(1)] = (i, int(m.group(2)))
        for k in ("Locations", "Curves", "Surfaces", "TShapes"):
            if k not in head:
                raise ValueError("not an ASCII BREP file (no %s table)" % k)
        self._locations(lines, head)
        self._curves(lines, head)
        self._surfaces(lines, head)
        self._tshapes(lines, head)
        self._walk()

    # -- tables --------------------------------------------------------------------------------
    def _locations(self, lines, head):
        i0, n = head["Locations"]
        i1 = head["Curve2ds"][0] if "Curve2ds" in head else head["Curves"][0]
        cur = _Cursor(" ".join(lines[i0 + 1:i1]).split())
        self.loc = [np.eye(4)]
        for _ in range(n):
            kind = int(cur.next())
            if kind == 1:
                T = np.eye(4)
                T[:3, :] = np.array(cur.floats(12)).reshape(3, 4)
            else:   # product of earlier locations raised to integer powers; the first listed factor acts first
                    # (checked on the reference's Embankment extrusion: "4 1 6 1" = rotate the sketch edge into
                    # the xz plane, then shift it by the extrusion vector)
                T = np.eye(4)
                while True:
                    k = int(cur.next())
                    if k == 0:
                        break
                    p = int(cur.next())
                    M = self.loc[k] if p > 0 else _inv(self.loc[k])
                    for _ in range(abs(p)):
                        T = M @ T
            self.loc.append(T)

    def _curves(self, lines, head):
        i0, n = head["Curves"]
        i1 = head["Polygon3D"][0] if "Polygon3D" in head else head["Surfaces"][0]
        cur = _Cursor(" ".join(lines[i0 + 1:i1]).split())
        self.curves = [None]
        try:
            for _ in range(n):
                self.curves.append(_read_curve(cur))
        except NotImplementedError:
            self.curves += [None] * (n + 1 - len(self.curves))
            self.curves_partial = True

    def _surfaces(self, lines, head):
        i0, n = head["Surfaces"]
        i1 = head["Triangulations"][0] if "Triangulations" in head else head["TShapes"][0]
        cur = _Cursor(" ".join(lines[i0 + 1:i1]).split())
        self.surfaces = [None]
        try:
            for _ in range(n):
                self.surfaces.append(_read_surface(cur))
        except NotImplementedError:
            self.surfaces += [None] * (n + 1 - len(self.surfaces))

    def _tshapes(self, lines, head):
        i, n = head["TShapes"]
        i += 1
        recs = []
        while len(recs) < n:
            while lines[i].strip() == "":
                i += 1
            kind = lines[i].strip()
            i += 1
            geo = []
            while not _FLAGS.match(lines[i].strip()):
                geo.append(lines[i])
                i += 1
            i += 1
            toks = []
            while True:
                t = lines[i].split()
                i += 1
                toks += t
                if t and t[-1] == "*":
                    break
            toks = toks[:-1]
            subs = [(1 if toks[k][0] != "-" else -1, abs(int(toks[k])), int(toks[k + 1])) for k in range(0, len(toks), 2)]
            recs.append((kind, [g for g in geo if g.strip() != ""], subs))
        self.n = n
        self.recs = recs        # file position p (0-based) holds shape number n - p
        rest = " ".join(lines[i:]).split()
        self.root = (abs(int(rest[0])), int(rest[1])) if len(rest) >= 2 else (1, 0)

    def _rec(self, num):
        return self.recs[self.n - num]

    # -- depth-first numbering -----------------------------------------------------------------
    def _walk(self):
        self.faces, self.edges, self.vertices = [], [], []
        seen = {}

        def key(num, T):
            return (num,) + tuple(np.round(T[:3, :].reshape(-1), 9))

        def visit(num, T):
            kind, geo, subs = self._rec(num)
            k = key(num, T)
            first = k not in seen
            seen[k] = True
            if first and kind == "Fa":
                self.faces.append((num, T))
            elif first and kind == "Ed":
                self.edges.append((num, T))
            elif first and kind == "Ve":
                self.vertices.append((num, T))
            if first or kind in ("Wi", "Sh", "So", "CS", "Co"):
                for _, sub, li in subs:
                    visit(sub, T @ self.loc[li])

        visit(self.root[0], self.loc[self.root[1]])

    # -- geometry of numbered sub-shapes (1-based, FreeCAD names) ---------------------------------
    @staticmethod
    def _pt(T, p):
        return T[:3, :3] @ p + T[:3, 3]

    def _vertex_point(self, num, T):
        _, geo, _ = self._rec(num)
        return self._pt(T, np.array([float(v) for v in geo[1].split()]))

    def _edge_polyline(self, num, T):
        _, geo, subs = self._rec(num)
        for g in geo[1:]:
            t = g.split()
            if len(t) == 5 and t[0] == "1":
                c = self.curves[int(t[1])]
                if c is None:
                    raise NotImplementedError("edge on a curve type that is not supported")
                Tc = T @ self.loc[int(t[2])]
                first, last = float(t[3]), float(t[4])
                if c[0] == "line":
                    pts = np.array([c[1] + first * c[2], c[1] + last * c[2]])
                else:
                    _, cen, _, xd, yd, r = c
                    m = max(2, int(np.ceil(abs(last - first) / (2.0 * np.pi) * self.arc_points * 4)))
                    tt = np.linspace(first, last, m + 1)
                    pts = cen + r * (np.outer(np.cos(tt), xd) + np.outer(np.sin(tt), yd))
                return pts @ Tc[:3, :3].T + Tc[:3, 3]
        # no 3D curve stored: straight between its vertices
        v = [self._vertex_point(s, T @ self.loc[li]) for _, s, li in subs]
        return np.array(v[:2])

    def vertex(self, n):
        if not 1 <= n <= len(self.vertices):
            raise ValueError("the shape has %d vertices, Vertex%d was asked for" % (len(self.vertices), n))
        s = SubShape("vertex", "Vertex%d" % n)
        s.point = self._vertex_point(*self.vertices[n - 1])
        return s

    def edge(self, n):
        if not 1 <= n <= len(self.edges):
            raise ValueError("the shape has %d edges, Edge%d was asked for" % (len(self.edges), n))
        s = SubShape("edge", "Edge%d" % n)
        p = self._edge_polyline(*self.edges[n - 1])
        s.segments = np.stack([p[:-1], p[1:]], 1)
        return s

    def face(self, n):
        if not 1 <= n <= len(self.faces):
            raise ValueError("the shape has %d faces, Face%d was asked for" % (len(self.faces), n))
        num, T = self.faces[n - 1]
        _, geo, subs = self._rec(num)
        t = geo[0].split()
        surf = self.surfaces[int(t[2])]
        Ts = T @ self.loc[int(t[3])]
        s = SubShape("face", "Face%d" % n)
        if surf is not None and surf[0] == "plane":
            nrm = Ts[:3, :3] @ surf[2]
            s.surface = ("plane", self._pt(Ts, surf[1]), nrm / np.linalg.norm(nrm))
        elif surf is not None:
            ax = Ts[:3, :3] @ surf[2]
            s.surface = ("cylinder", self._pt(Ts, surf[1]), ax / np.linalg.norm(ax), surf[3])
        segs = []
        for wsign, wnum, wli in subs:
            Tw = T @ self.loc[wli]
            for esign, enum, eli in self._rec(wnum)[2]:
                p = self._edge_polyline(enum, Tw @ self.loc[eli])
                if wsign * esign < 0:
                    p = p[::-1]
                segs.append(np.stack([p[:-1], p[1:]], 1))
        s.segments = np.concatenate(segs, 0)
        return s

    def element(self, name):
        m = re.match(r"^(Face|Edge|Vertex)(\d+)$", name)
        if not m:
            raise ValueError("not a sub-shape name: %r" % name)
        return {"Face": self.face, "Edge": self.edge, "Vertex": self.vertex}[m.group(1)](int(m.group(2)))
