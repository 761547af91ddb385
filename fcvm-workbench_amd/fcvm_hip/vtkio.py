"""Legacy-VTK (DataFile Version 5.1, BINARY) unstructured-grid reader / writer.

The reference exports its results with ``pv.save_meshio(file, grid)``
(reference: source code/fcVM.py:2903-2950, exportVTK), which lands in meshio's
legacy VTK 5.1 binary writer.  pyvista / meshio are not part of this framework,
so the same on-disk layout is produced (and parsed back) here:

    # vtk DataFile Version 5.1
    written by <producer>
    BINARY
    DATASET UNSTRUCTURED_GRID
    POINTS <n> double                 big-endian f64 [n,3]
    CELLS <ncell+1> <nconn>
    OFFSETS vtktypeint64              big-endian i64 [ncell+1]
    CONNECTIVITY vtktypeint64         big-endian i64 [nconn]
    CELL_TYPES <ncell>                big-endian i32 [ncell]   (24 = quadratic tetra)
    POINT_DATA <n>
    FIELD FieldData <k>
    <name> <ncomp> <n> double         big-endian f64 [n,ncomp]   (k times)

Field names: blanks become underscores, a trailing new-line in the name (the
reference's scalar names carry one, e.g. "Critical Strain Ratio\\n") is kept.
"""
import re

import numpy as np

VTK_QUADRATIC_TETRA = 24

_DT = {"double": ">f8", "float": ">f4", "int": ">i4", "vtktypeint64": ">i8", "long": ">i8"}


def write_vtk(path, points, cells, point_data, producer="fcvm_hip"):
    """points [n,3] f64, cells [ne,10] 0-based ints, point_data: ordered dict name -> [n] or [n,c]."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    cells = np.ascontiguousarray(cells, dtype=np.int64)
    n, ne, npc = len(points), len(cells), cells.shape[1]
    with open(path, "wb") as f:
        f.write(b"# vtk DataFile Version 5.1\n")
        f.write(("written by %s\n" % producer).encode())
        f.write(b"BINARY\nDATASET UNSTRUCTURED_GRID\n")
        f.write(("POINTS %d double\n" % n).encode())
        f.write(points.astype(">f8").tobytes())
        f.write(b"\n")
        f.write(("CELLS %d %d\n" % (ne + 1, ne * npc)).encode())
        f.write(b"OFFSETS vtktypeint64\n")
        f.write((np.arange(ne + 1, dtype=np.int64) * npc).astype(">i8").tobytes())
        f.write(b"\n")
        f.write(b"CONNECTIVITY vtktypeint64\n")
        f.write(cells.astype(">i8").tobytes())
        f.write(b"\n")
        f.write(("CELL_TYPES %d\n" % ne).encode())
        f.write(np.full(ne, VTK_QUADRATIC_TETRA, dtype=">i4").tobytes())
        f.write(b"\n")
        f.write(("POINT_DATA %d\n" % n).encode())
        f.write(("FIELD FieldData %d\n" % len(point_data)).encode())
        for name, arr in point_data.items():
            arr = np.asarray(arr, dtype=np.float64)
            if arr.ndim == 1:
                arr = arr.reshape(n, 1)
            f.write(("%s %d %d double\n" % (name.replace(" ", "_"), arr.shape[1], n)).encode())
            f.write(np.ascontiguousarray(arr).astype(">f8").tobytes())
            f.write(b"\n")


def read_vtk(path):
    """Returns (points [n,3], cells [ne,npc] 0-based, point_data dict)."""
    buf = open(path, "rb").read()
    pos = 0

    def line():
        nonlocal pos
        e = buf.index(b"\n", pos)
        s = buf[pos:e]
        pos = e + 1
        return s

    def skipnl():
        nonlocal pos
        while pos < len(buf) and buf[pos:pos + 1] == b"\n":
            pos += 1

    def block(count, dt):
        nonlocal pos
        a = np.frombuffer(buf, dtype=np.dtype(dt), count=count, offset=pos)
        pos += a.nbytes
        return a

    assert line().startswith(b"# vtk DataFile Version")
    line()
    assert line().strip() == b"BINARY"
    assert line().strip() == b"DATASET UNSTRUCTURED_GRID"
    t = line().split()
    assert t[0] == b"POINTS"
    n = int(t[1])
    points = block(3 * n, _DT[t[2].decode()]).astype(np.float64).reshape(n, 3)
    skipnl()
    t = line().split()
    assert t[0] == b"CELLS"
    noff, nconn = int(t[1]), int(t[2])
    t = line().split()
    assert t[0] == b"OFFSETS"
    off = block(noff, _DT[t[1].decode()]).astype(np.int64)
    skipnl()
    t = line().split()
    assert t[0] == b"CONNECTIVITY"
    conn = block(nconn, _DT[t[1].decode()]).astype(np.int64)
    skipnl()
    t = line().split()
    assert t[0] == b"CELL_TYPES"
    ne = int(t[1])
    block(ne, ">i4")
    skipnl()
    npc = int(off[1] - off[0])
    assert np.all(np.diff(off) == npc)
    cells = conn.reshape(ne, npc)
    data = {}
    t = line().split()
    assert t[0] == b"POINT_DATA" and int(t[1]) == n
    t = line().split()
    assert t[0] == b"FIELD"
    k = int(t[2])
    hdr = re.compile(rb"([^\x00]*?)\s(\d+) (\d+) (double|float|int)\n", re.S)
    for _ in range(k):
        m = hdr.match(buf, pos)
        assert m, "bad field header at %d" % pos
        name = m.group(1).decode()
        nc, nt, dt = int(m.group(2)), int(m.group(3)), m.group(4).decode()
        pos = m.end()
        arr = block(nc * nt, _DT[dt]).astype(np.float64).reshape(nt, nc)
        skipnl()
        data[name] = arr[:, 0].copy() if nc == 1 else arr
    return points, cells, data
