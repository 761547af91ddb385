"""Example problems built from the reference's own models (data under tests/golden/, extracted from
the reference's result / control / FreeCAD files by tests/golden/make_golden.py)."""
import json
import os

import numpy as np

from . import inpout, meshgen

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


def embankment(refine=0):
    """The reference's Embankment_with_Ditch_Example (659 tet10, gravity-driven collapse, nu = 0.45),
    optionally refined uniformly `refine` times (x8 elements each).  Boundary conditions as in the
    FreeCAD document: bottom and the two x-ends clamped, the two extrusion caps fixed in y.
    Returns dict(elNodes, nocoord, material, fix, ctl, name)."""
    g = np.load(os.path.join(_GOLDEN, "embankment.npz"))
    meta = json.load(open(os.path.join(_GOLDEN, "embankment.json")))
    el, xyz = g["elNodes"], g["nocoord"]
    lo, hi = xyz.min(0), xyz.max(0)
    for _ in range(refine):
        el, xyz = meshgen.refine_tet10(el, xyz)
    tol = 1e-6
    on = lambda ax, v: np.abs(xyz[:, ax] - v) < tol  # noqa: E731
    fix = {}
    for n in np.where(on(0, lo[0]) | on(0, hi[0]) | on(2, lo[2]))[0]:
        for k in range(3):
            fix[3 * int(n) + k] = 0.0
    for n in np.where(on(1, lo[1]) | on(1, hi[1]))[0]:
        fix[3 * int(n) + 1] = 0.0
    return dict(elNodes=el, nocoord=xyz, material=g["material"], fix=fix, ctl=inpout.parse_inp_lines(meta["inp"]),
                name=meta["name"] + ("_refined%d" % refine if refine else ""))


PLATE_WITH_HOLE_INP = ("100", "0.0", "0.0", "0.0", "10", "20", "5.0e-3", "1.2", "2.0", "1.2", "1.2", "total", "0.25", "0.0", "1.5",
                       "PEEQ", "unaveraged", "GNLN", "10", "1.0", "0.0")


def plate_with_hole(h=10.0):
    """The reference's Plate_with_hole_Example (BASELINE configs[1] and the north-star workload) on a
    synthetic mesh of edge length ~h: its FreeCAD document carries geometry, constraints and material but
    no mesh (FemMesh.unv is empty).  From Document.xml: Cut = Box(200 x 100 x 10) - Cylinder(R 50, axis z
    through the origin corner); ConstraintDisplacement x on Face1 (x = 0), y on Face7 (x = 200),
    z on Vertex1 / 2 / 5 (resolved with fcvm_hip.brep from the document's PartShape2.brp, tests/test_host.py); ConstraintPressure 100 N/mm^2 (reversed = pulling) on the face x = 200; steel
    (E 210000, nu 0.3, 7900 kg/m^3); control file values as shipped (sig_yield 100, 10 arc-length steps).
    The load pulls eccentrically on the ligament x = 0, 50 <= y <= 100 (tension + in-plane bending), which
    is why the reference's .out plateaus at a load factor of about 0.23 and not at the net-section 0.5.
    The pressure is applied as the equivalent uniform traction (identical for the geometrically linear run).
    Returns dict(elNodes, nocoord, material, fix, ufaces, uloads, ctl, name)."""
    el, xyz = meshgen.plate_with_hole(h)
    fix = {}
    for n in meshgen.nodes_on_plane(xyz, 0, 0.0, tol=1e-7):
        fix[3 * int(n)] = 0.0
    for n in meshgen.nodes_on_plane(xyz, 0, 200.0, tol=1e-7):
        fix[3 * int(n) + 1] = 0.0
    for p in ((0.0, 50.0, 10.0), (0.0, 50.0, 0.0), (50.0, 0.0, 0.0)):   # Vertex1, Vertex2, Vertex5 of the Cut
        n = int(np.argmin(np.abs(xyz - np.array(p)).max(1)))
        assert np.abs(xyz[n] - p).max() < 1e-6
        fix[3 * n + 2] = 0.0
    faces = meshgen.boundary_faces_on_plane(el, xyz, 0, 200.0, tol=1e-7)
    return dict(elNodes=el, nocoord=xyz, material=np.array([210000.0, 0.3, 7.9e-6]), fix=fix, ufaces=faces,
                uloads=np.tile([100.0, 0.0, 0.0], (len(faces), 1)), ctl=inpout.parse_inp_lines(PLATE_WITH_HOLE_INP),
                name="Plate_with_hole_Example_h%g" % h)
