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
