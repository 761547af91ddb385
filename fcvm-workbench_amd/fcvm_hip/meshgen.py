"""Synthetic tet10 meshes in fcVM's internal node order (BASELINE configs: "synthetic cube").

Node order inside an element follows the reference's shape functions (fcVM.py:365-380): corners
0-3, then mid-side nodes 4:(0,1) 5:(1,2) 6:(0,2) 7:(0,3) 8:(1,3) 9:(2,3).
"""
import itertools

import numpy as np

# local (corner, corner) pair of each mid-side node
MID_PAIRS = ((0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3))
# the four 6-node faces of a tet10 in the 6-node triangle order of shape6tri (fcVM.py:485-496):
# corners 0,1,2 then mids (0,1), (1,2), (0,2)
TET_FACES = ((0, 1, 2, 4, 5, 6), (0, 1, 3, 4, 8, 7), (1, 2, 3, 5, 9, 8), (0, 2, 3, 6, 9, 7))


def cube_tet10(nx, ny, nz, lx=1.0, ly=1.0, lz=1.0):
    """Box [0,lx]x[0,ly]x[0,lz] cut into nx*ny*nz cells of 6 Kuhn tetrahedra each (conforming).

    Returns (elNodes int64 [6*nx*ny*nz, 10] 1-based, nocoord f64 [(2nx+1)(2ny+1)(2nz+1), 3]).
    Nodes live on the half-cell lattice, numbered x-fastest."""
    mx, my, mz = 2 * nx + 1, 2 * ny + 1, 2 * nz + 1
    gi, gj, gk = np.meshgrid(np.arange(mx), np.arange(my), np.arange(mz), indexing="ij")
    nocoord = np.empty((mx * my * mz, 3))
    idx = (gk * my + gj) * mx + gi
    nocoord[idx.ravel(), 0] = gi.ravel() * (lx / (2 * nx))
    nocoord[idx.ravel(), 1] = gj.ravel() * (ly / (2 * ny))
    nocoord[idx.ravel(), 2] = gk.ravel() * (lz / (2 * nz))
    ci, cj, ck = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    # lattice number of every cell's origin; a node of the cell at lattice offset (ox, oy, oz) is that number + (oz * my + oy) * mx + ox
    base_id = ((2 * ck.ravel()) * my + 2 * cj.ravel()) * mx + 2 * ci.ravel() + 1
    offs = np.empty((6, 10), dtype=np.int64)
    for t, perm in enumerate(itertools.permutations(range(3))):
        v = [np.zeros(3, dtype=np.int64)]
        for ax in perm:
            s = v[-1].copy()
            s[ax] += 2
            v.append(s)
        v = np.array(v)  # 4 corner offsets on the lattice
        d = np.linalg.det((v[1:] - v[0]).astype(float))
        if d < 0:
            v[[1, 2]] = v[[2, 1]]
        loc = [v[0], v[1], v[2], v[3]] + [(v[a] + v[b]) // 2 for a, b in MID_PAIRS]
        offs[t] = [(o[2] * my + o[1]) * mx + o[0] for o in loc]
    elNodes = (base_id[:, None, None] + offs[None, :, :]).reshape(-1, 10)
    return elNodes, nocoord


def boundary_faces_on_plane(elNodes, nocoord, axis, value, tol=1e-9):
    """6-node faces (1-based node numbers, shape6tri order) of elements lying in the plane
    x[axis] == value."""
    on = np.abs(nocoord[:, axis] - value) < tol
    out = []
    for f in TET_FACES:
        nd = elNodes[:, list(f)]
        m = on[nd[:, 0] - 1] & on[nd[:, 1] - 1] & on[nd[:, 2] - 1]
        out.append(nd[m])
    return np.concatenate(out, axis=0)


def nodes_on_plane(nocoord, axis, value, tol=1e-9):
    return np.where(np.abs(nocoord[:, axis] - value) < tol)[0]


def cube_case(n, l=100.0, traction=(10.0, 0.0, 100.0)):
    """The benchmark / test problem: n^3 cells, bottom face clamped, uniform traction on the top.

    Returns dict(elNodes, nocoord, fix, loads kwargs)."""
    elNodes, nocoord = cube_tet10(n, n, n, l, l, l)
    fix = {}
    for nd in nodes_on_plane(nocoord, 2, 0.0):
        for k in range(3):
            fix[3 * int(nd) + k] = 0.0
    faces = boundary_faces_on_plane(elNodes, nocoord, 2, l)
    uloads = np.tile(np.asarray(traction, dtype=np.float64), (len(faces), 1))
    return dict(elNodes=elNodes, nocoord=nocoord, fix=fix, ufaces=faces, uloads=uloads)


def refine_tet10(elNodes, nocoord):
    """Uniform (red) refinement of a straight-sided tet10 mesh: every element is cut into 8 children
    whose corners are the parent's 10 nodes (4 corner tets + the inner octahedron split along the
    diagonal m02-m13, Bey's rule); new nodes are the midpoints of the children's edges.  Existing
    nodes keep their numbers, new nodes are appended.  Returns (elNodes2 [8*ne,10], nocoord2).
    Used to scale the reference's example meshes (BASELINE config "Embankment refined to ~500k")."""
    el = np.asarray(elNodes, dtype=np.int64) - 1
    xyz = np.asarray(nocoord, dtype=np.float64)
    v0, v1, v2, v3, m01, m12, m02, m03, m13, m23 = (el[:, k] for k in range(10))
    kids = [(v0, m01, m02, m03), (m01, v1, m12, m13), (m02, m12, v2, m23), (m03, m13, m23, v3),
            (m01, m02, m03, m13), (m01, m02, m12, m13), (m02, m03, m13, m23), (m02, m12, m13, m23)]
    c = np.stack([np.stack(k, axis=1) for k in kids], axis=1).reshape(-1, 4)  # [8ne,4], children of a parent adjacent
    e1, e2, e3 = xyz[c[:, 1]] - xyz[c[:, 0]], xyz[c[:, 2]] - xyz[c[:, 0]], xyz[c[:, 3]] - xyz[c[:, 0]]
    neg = np.einsum("ij,ij->i", e1, np.cross(e2, e3)) < 0
    c[neg, 1], c[neg, 2] = c[neg, 2].copy(), c[neg, 1].copy()
    pairs = np.stack([c[:, [a, b]] for a, b in MID_PAIRS], axis=1).reshape(-1, 2)  # [8ne*6,2]
    pairs.sort(axis=1)
    key = pairs[:, 0] * (len(xyz) + 1) + pairs[:, 1]
    uniq, inv = np.unique(key, return_inverse=True)
    a, b = uniq // (len(xyz) + 1), uniq % (len(xyz) + 1)
    new_xyz = 0.5 * (xyz[a] + xyz[b])
    mids = (len(xyz) + inv).reshape(-1, 6)
    el2 = np.concatenate([c, mids], axis=1) + 1
    return el2.astype(np.int64), np.concatenate([xyz, new_xyz], axis=0)


def fix_orientation(elNodes, nocoord):
    """Swaps corners 1 <-> 2 (and the mid-side nodes that go with them) of every element whose corner
    Jacobian is negative.  In place on elNodes; returns the number of elements flipped."""
    el = elNodes - 1
    p = nocoord
    e1, e2, e3 = p[el[:, 1]] - p[el[:, 0]], p[el[:, 2]] - p[el[:, 0]], p[el[:, 3]] - p[el[:, 0]]
    neg = np.einsum("ij,ij->i", e1, np.cross(e2, e3)) < 0
    for a, b in ((1, 2), (4, 6), (8, 9)):  # c1<->c2 : m01<->m02, m13<->m23; m12 and m03 stay
        elNodes[neg, a], elNodes[neg, b] = elNodes[neg, b].copy(), elNodes[neg, a].copy()
    return int(neg.sum())


def merge_blocks(blocks, tol=1e-7, lattices=None):
    """Glues structured blocks [(elNodes, nocoord), ...] with matching interface lattices into one mesh:
    nodes closer than tol (per coordinate, after rounding) become one.  Straight-sided elements: every
    mid-side node is put at the mean of its two corners before merging, so mapped blocks stay conforming.
    lattices (optional): per block the half-cell lattice (mx, my, mz) of a cube_tet10 block whose nodes were only moved -- the
    fast path of the large meshes (8 M tet10 for the 8-GPU bench: 64 -> 30 s): a mid-side node of a Kuhn triangulation sits between the
    lattice points p - o and p + o (o = the odd coordinates of p), so the corners are found per NODE instead of per element edge, and
    only nodes on a block's boundary faces can coincide with another block's, so only those are compared.  Same mesh, same numbering."""
    els, pts, cand, off = [], [], [], 0
    for nb, (el, xyz) in enumerate(blocks):
        xyz = xyz.copy()
        lat = lattices[nb] if lattices is not None else None
        if lat is not None:
            mx, my, mz = lat
            assert len(xyz) == mx * my * mz
            idx = np.arange(len(xyz), dtype=np.int64)
            i, j, k = idx % mx, (idx // mx) % my, idx // (mx * my)
            oi, oj, ok = i & 1, j & 1, k & 1
            mid = np.nonzero((oi | oj | ok) > 0)[0]
            d = ((ok * my + oj) * mx + oi)[mid]
            xyz[mid] = 0.5 * (xyz[mid - d] + xyz[mid + d])
            onb = (i == 0) | (i == mx - 1) | (j == 0) | (j == my - 1) | (k == 0) | (k == mz - 1)
            cand.append(np.nonzero(onb)[0] + off)
        else:
            e0 = el - 1
            for k, (a, b) in enumerate(MID_PAIRS):
                xyz[e0[:, 4 + k]] = 0.5 * (xyz[e0[:, a]] + xyz[e0[:, b]])
            cand.append(np.arange(len(xyz), dtype=np.int64) + off)
        els.append(el + off)
        pts.append(xyz)
        off += len(xyz)
    el, xyz = np.concatenate(els), np.concatenate(pts)
    cand = np.concatenate(cand)
    key = np.round(xyz[cand] / tol).astype(np.int64)
    # one 64-bit hash per lattice point instead of a row-wise unique (5x faster at 1.5 M nodes); a collision between
    # different points would merge them, so it is checked for and falls back to the exact row-wise path
    h = (key[:, 0] * np.int64(73856093)) ^ (key[:, 1] * np.int64(19349663)) ^ (key[:, 2] * np.int64(83492791))
    _, first, inv = np.unique(h, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    if not np.array_equal(key[first[inv]], key):
        _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
        inv = inv.reshape(-1)
    # every node's representative = its first appearance (block by block, lattice order inside a block); the kept nodes are
    # numbered in that order
    rep = np.arange(len(xyz), dtype=np.int64)
    rep[cand] = cand[first[inv]]
    keep = rep == np.arange(len(xyz))
    newid = np.cumsum(keep) - 1
    el = newid[rep[el - 1]] + 1
    xyz = xyz[keep]
    fix_orientation(el, xyz)
    return el.astype(np.int64), xyz


def plate_with_hole(h, length=200.0, width=100.0, thick=10.0, radius=50.0):
    """The geometry of the reference's Plate_with_hole_Example (FreeCAD document: Part::Box 200 x 100 x 10
    minus a Part::Cylinder of radius 50 about the z axis through the box corner at the origin), meshed with
    mapped Kuhn tetrahedra; h = edge length away from the hole.  Two blocks: a strip between the quarter arc
    and the polyline (w,0) -> (w,w) -> (0,w) (w = width), parameters radial x angular x z with rays from the
    origin and radii in geometric progression (dr / r = d theta: near-isotropic cells that shrink towards the
    hole, where the plastic zone starts), and the box [w, length] x [0, w] behind it.  The thickness is
    divided to the geometric mean of the in-plane sizes.  Returns (elNodes, nocoord)."""
    w = width
    n_a = max(2, int(round(w / h)))
    dth = 0.25 * np.pi / n_a
    n_t = max(2, int(round(0.5 * (np.log(w / radius) + np.log(np.sqrt(2.0) * w / radius)) / dth)))
    n_x = max(1, int(round((length - w) / h)))
    n_z = max(2, int(round(thick / (np.sqrt(radius * dth * h)))))
    # strip: parameter box (t radial, s along the arc, z); s in [0,1] -> outer point on x = w, s in [1,2] -> on y = w
    el1, q = cube_tet10(n_t, 2 * n_a, n_z, 1.0, 2.0, thick)
    t, s = q[:, 0], q[:, 1]
    out = np.where((s <= 1.0)[:, None], np.stack([np.full_like(s, w), w * s], axis=1),
                   np.stack([w * (2.0 - s), np.full_like(s, w)], axis=1))   # uniform points on the polyline
    rout = np.linalg.norm(out, axis=1)
    r = radius * (rout / radius) ** t
    p1 = np.concatenate([out * (r / rout)[:, None], q[:, 2:3]], axis=1)
    el2, p2 = cube_tet10(n_x, n_a, n_z, length - w, w, thick)
    p2[:, 0] += w
    lat = lambda a, b, c: (2 * a + 1, 2 * b + 1, 2 * c + 1)  # noqa: E731
    return merge_blocks([(el1, p1), (el2, p2)], lattices=[lat(n_t, 2 * n_a, n_z), lat(n_x, n_a, n_z)])
