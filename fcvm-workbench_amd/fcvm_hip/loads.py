"""External (surface / edge / vertex) load vector -- host-side, O(surface) work.

Mirrors the load loops at the top of calcGSM / calcTSM (reference fcVM.py:649-727, 858-938):
pressure on 6-node faces (follower pressure when `disp` is given), point forces, uniformly
distributed face tractions and edge line loads.  Gravity is integrated on the GPU together with
the element stiffness (fcvm_assemble).
"""
import numpy as np

# fcVM.py:598-611
GP6 = np.array([[0.445948490915965, 0.445948490915965, 0.111690794839005],
                [0.10810301816807, 0.445948490915965, 0.111690794839005],
                [0.445948490915965, 0.10810301816807, 0.111690794839005],
                [0.091576213509771, 0.091576213509771, 0.054975871827661],
                [0.816847572980458, 0.091576213509771, 0.054975871827661],
                [0.091576213509771, 0.816847572980458, 0.054975871827661]])
GP2 = np.array([[-0.5773502691896257, 1.0], [0.5773502691896257, 1.0]])


class Loads:
    """Load entities with REAL rows only (the reference's dummy numba-signature rows,
    fcVM.py:260-267, are dropped / re-added at the fcVM.py API mirror)."""

    def __init__(self, pfaces=None, pvals=None, vnodes=None, vloads=None, ufaces=None, uloads=None,
                 edges=None, eloads=None):
        def arr(a, shape, dt):
            return np.zeros(shape, dt) if a is None else np.ascontiguousarray(a, dt).reshape((-1,) + shape[1:])
        self.pfaces = arr(pfaces, (0, 6), np.int64)
        self.pvals = arr(pvals, (0,), np.float64)
        self.vnodes = arr(vnodes, (0,), np.int64)
        self.vloads = arr(vloads, (0, 3), np.float64)
        self.ufaces = arr(ufaces, (0, 6), np.int64)
        self.uloads = arr(uloads, (0, 3), np.float64)
        self.edges = arr(edges, (0, 3), np.int64)
        self.eloads = arr(eloads, (0, 3), np.float64)

    @property
    def has_pressure(self):
        return len(self.pfaces) > 0


def _tri6(xi, et):
    """shape6tri (fcVM.py:485-512): shape functions and local derivatives at (xi, et)."""
    shp = np.array([(1.0 - xi - et) * (1.0 - 2.0 * xi - 2.0 * et), xi * (2.0 * xi - 1.0), et * (2.0 * et - 1.0),
                    4.0 * xi * (1.0 - xi - et), 4.0 * xi * et, 4.0 * et * (1 - xi - et)])
    d = np.array([[-3.0 + 4.0 * et + 4.0 * xi, -1.0 + 4.0 * xi, 0.0, -4.0 * (-1.0 + et + 2.0 * xi), 4.0 * et, -4.0 * et],
                  [-3.0 + 4.0 * et + 4.0 * xi, 0.0, -1.0 + 4.0 * et, -4.0 * xi, 4.0 * xi, -4.0 * (-1.0 + 2.0 * et + xi)]])
    return shp, d


def _face_integrals(x6):
    """x6 [nf,6,3] -> per Gauss point: shp [6], |J| [nf], unit normal [nf,3], weight."""
    for xi, et, w in GP6:
        shp, d = _tri6(xi, et)
        t0 = np.einsum("k,fkc->fc", d[0], x6)
        t1 = np.einsum("k,fkc->fc", d[1], x6)
        xp = np.cross(t0, t1)
        xsj = np.linalg.norm(xp, axis=1)
        yield shp, xsj, xp / xsj[:, None], w


def external_loads(nocoord, loads, disp=None):
    """Returns glv [3*nn] (without gravity)."""
    nocoord = np.asarray(nocoord, dtype=np.float64)
    glv = np.zeros(nocoord.size)
    g3 = glv.reshape(-1, 3)
    if len(loads.pfaces):  # fcVM.py:649-672 / 858-883
        x = nocoord if disp is None else nocoord + np.asarray(disp).reshape(-1, 3)
        x6 = x[loads.pfaces - 1]
        for shp, xsj, xp, w in _face_integrals(x6):
            f = (loads.pvals * np.abs(xsj) * w)[:, None, None] * shp[None, :, None] * xp[:, None, :]
            np.add.at(g3, loads.pfaces - 1, f)
    if len(loads.vnodes):  # fcVM.py:674-681
        np.add.at(g3, loads.vnodes - 1, loads.vloads)
    if len(loads.ufaces):  # fcVM.py:683-705
        x6 = nocoord[loads.ufaces - 1]
        for shp, xsj, xp, w in _face_integrals(x6):
            f = (np.abs(xsj) * w)[:, None, None] * shp[None, :, None] * loads.uloads[:, None, :]
            np.add.at(g3, loads.ufaces - 1, f)
    if len(loads.edges):  # fcVM.py:707-727 with shape2lin fcVM.py:545-565
        x3 = nocoord[loads.edges - 1]
        for xi, w in GP2:
            shp = np.array([-0.5 * (1.0 - xi) * xi, 0.5 * (1.0 + xi) * xi, (1.0 + xi) * (1.0 - xi)])
            d = np.array([xi - 0.5, xi + 0.5, -2.0 * xi])
            xsj = np.linalg.norm(np.einsum("k,ekc->ec", d, x3), axis=1)
            f = (np.abs(xsj) * w)[:, None, None] * shp[None, :, None] * loads.eloads[:, None, :]
            np.add.at(g3, loads.edges - 1, f)
    return glv


def split_follower(loads):
    """(the loads that do not follow the geometry, the pressure faces alone) -- calcTSM re-integrates only the pressure on the deformed
    geometry (fcVM.py:858-883); everything else is the same vector at every tangent"""
    static = Loads(vnodes=loads.vnodes, vloads=loads.vloads, ufaces=loads.ufaces, uloads=loads.uloads, edges=loads.edges, eloads=loads.eloads)
    return static, Loads(pfaces=loads.pfaces, pvals=loads.pvals)


class FollowerPressure:
    """The pressure part of the external load vector on the deformed geometry from the displacements of the loaded faces' nodes ALONE:
    nodes = their (0-based, sorted, distinct) numbers; loads(x_nodes) -> [len(nodes), 3] nodal forces at those nodes (the face integrals of
    external_loads restricted to that node set).  The solver gathers disp[nodes] on the device, calls this on the host (O(loaded faces)) and
    scatters the result back: no full displacement vector leaves the GPU per GNLY tangent."""

    def __init__(self, nocoord, loads):
        self.nodes = np.unique(loads.pfaces.reshape(-1) - 1)
        self.faces = np.searchsorted(self.nodes, loads.pfaces - 1)      # local numbering within `nodes`
        self.pvals = loads.pvals
        self.x0 = np.ascontiguousarray(np.asarray(nocoord, dtype=np.float64)[self.nodes])

    def loads(self, disp_nodes):
        x6 = (self.x0 + np.asarray(disp_nodes, dtype=np.float64).reshape(-1, 3))[self.faces]
        g3 = np.zeros((len(self.nodes), 3))
        for shp, xsj, xp, w in _face_integrals(x6):
            f = (self.pvals * np.abs(xsj) * w)[:, None, None] * shp[None, :, None] * xp[:, None, :]
            np.add.at(g3, self.faces, f)
        return g3
