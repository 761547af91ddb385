"""fcVM -- MI355X-native drop-in for the solver path of the reference's ``source code/fcVM.py``.

Same function names, argument order / meaning and return tuples as the reference for the hot
path, so that ``fcVM.FCMacro`` (reference: source code/fcVM.FCMacro:118-175) can call
``ft.calcGSM`` / ``ft.calcDisp`` / ``ft.mapStresses`` / ``ft.exportVTK`` unchanged; the work runs
in hand-written HIP kernels (libfcvm_hip.so) on the GPU and raises FcvmError if that library or a
GPU is missing -- there is no CPU fallback.

Not mirrored here (out of scope): the FreeCAD document access (setUpAnalysis,
setUpInput, pasteResults), the interactive matplotlib / pyvista windows (plot) and the
multi-GPU form of the "GNLY" buckling pre-analysis.

Usage like the reference macro:   sys.path.append(<repo>/fcvm-workbench_amd); import fcVM as ft
"""
import os

import numpy as np

from fcvm_hip import inpout, vtkio
from fcvm_hip._lib import FcvmError  # noqa: F401
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel, calc_disp

settings = {"solver": "hip-pcg", "rtol": 1.0e-10, "device": 0}


class HeadlessWindow:
    """Stand-in for the Qt widget the reference updates (fcVM.py:1104-1108): records the values."""

    class _W:
        def __init__(self):
            self.value = None

        def setValue(self, v):
            self.value = v

        def setText(self, v):
            self.value = v

        def isChecked(self):
            return bool(self.value)

    def __init__(self):
        self.progressBar, self.Step, self.Load_Factor = self._W(), self._W(), self._W()
        self.PEEQ, self.CSR, self.csrRbtn = self._W(), self._W(), self._W()


class DeviceStiffness:
    """What calcGSM returns in place of the COO value array `stm`: the assembled global
    stiffness matrix lives in HBM.  `to_coo()` gives the reference's lower-triangular COO."""

    def __init__(self, model, glv, modf, xgp):
        self.model, self.glv_dev, self.modf_dev, self.xgp_dev = model, glv, modf, xgp

    def to_coo(self):
        r, c, v = self.model.ctx.matrix_coo()
        m = r >= c
        return v[m], r[m], c[m]

    def __len__(self):
        return 9 * self.model.ctx.pattern_info()["nnz_blocks"]


def _loads(loadfaces, pressure, loadvertices, vertexloads, loadedges, edgeloads, loadfaces_uni, faceloads):
    # the first row of every load array is the reference's numba-signature dummy (fcVM.py:260-267)
    return Loads(pfaces=np.asarray(loadfaces)[1:], pvals=np.asarray(pressure)[1:],
                 vnodes=np.asarray(loadvertices)[1:, 0] if len(loadvertices) > 1 else None,
                 vloads=np.asarray(vertexloads)[1:], ufaces=np.asarray(loadfaces_uni)[1:],
                 uloads=np.asarray(faceloads)[1:], edges=np.asarray(loadedges)[1:], eloads=np.asarray(edgeloads)[1:])


def gaussPoints():
    """fcVM.py:587-613"""
    from fcvm_hip.loads import GP2, GP6
    a, b, w = 0.138196601125011, 0.585410196624968, 0.041666666666667
    gp10 = np.array([[a, a, a, w], [b, a, a, w], [a, b, a, w], [a, a, b, w]])
    return gp10, GP6.copy(), GP2.copy()


def calcGSM(elNodes, nocoord, materialbyElement, fix, grav_x, grav_y, grav_z, loadfaces, pressure, loadvertices,
            vertexloads, loadedges, edgeloads, loadfaces_uni, faceloads):
    """fcVM.py:621-816.  Returns (stm, row, col, glv, modf, V, loadsumx, loadsumy, loadsumz, ne, nn, x)
    with stm a DeviceStiffness handle (row / col are empty: the matrix stays on the GPU)."""
    loads = _loads(loadfaces, pressure, loadvertices, vertexloads, loadedges, edgeloads, loadfaces_uni, faceloads)
    model = GpuModel(elNodes, nocoord, np.asarray(materialbyElement)[0], fix, loads, device=settings["device"])
    glv, modf, V, xgp = model.calc_gsm((grav_x, grav_y, grav_z))
    glv_h = glv.get()
    g3 = glv_h.reshape(-1, 3)
    stm = DeviceStiffness(model, glv, modf, xgp)
    return (stm, np.zeros(0, np.int64), np.zeros(0, np.int64), glv_h, modf.get(), V, g3[:, 0].sum(), g3[:, 1].sum(),
            g3[:, 2].sum(), model.ne, model.nn, xgp.get().reshape(-1, 3))


def calcDisp(elNodes, nocoord, fixdof, movdof, modf, materialbyElement, stm, row, col, glv, nstep, iterat_max,
             error_max, relax, scale_re, scale_up, scale_dn, sig_yield_inp, disp_output, ultimate_strain, fcVM_window,
             Et_E, target_LF, x, noce, fix, grav_x, grav_y, grav_z, loadfaces, pload, loadvertices, vertexloads,
             loadedges, edgeloads, loadfaces_uni, faceloads, gnl, maxImp, ev1, ev2):
    """fcVM.py:1083-1635: same arguments, same 19-tuple result."""
    if isinstance(stm, DeviceStiffness):
        model = stm.model
    else:
        loads = _loads(loadfaces, pload, loadvertices, vertexloads, loadedges, edgeloads, loadfaces_uni, faceloads)
        model = GpuModel(elNodes, nocoord, np.asarray(materialbyElement)[0], fix, loads, device=settings["device"])
    ctl = dict(nstep=nstep, iterat_max=iterat_max, error_max=error_max, relax=relax, scale_re=scale_re,
               scale_up=scale_up, scale_dn=scale_dn, sig_yield=sig_yield_inp, disp_output=disp_output,
               ultimate_strain=ultimate_strain, Et_E=Et_E, target_LF=target_LF, grav_x=grav_x, grav_y=grav_y,
               grav_z=grav_z, gnl=gnl, maxImp=maxImp, ev1=ev1, ev2=ev2)
    w = fcVM_window

    def log(msg):
        if w is not None and msg.startswith("Step: "):
            w.Step.setText(msg[6:])
        print(msg)

    res = calc_disp(model, ctl, log=log, rtol=settings["rtol"])
    if w is not None:
        w.Load_Factor.setText(str(round(float(res["lbd"][-1]), 3)))
    g = lambda k: res[k].get()  # noqa: E731
    eigenval, eigenvec = res["eigenval"], res["eigenvec"]
    if hasattr(nocoord, "__setitem__") and res["nocoord"].shape == np.shape(nocoord):
        nocoord[:] = res["nocoord"]  # the reference adds the imperfection to nocoord in place (fcVM.py:1240)
    return (res["displacements"], res["disp_el"], eigenval, eigenvec, g("stresses"), g("peeq"), g("sigmises"),
            g("csr"), res["lout"], list(res["un"]), list(res["crip"]), list(res["peeqplot"]), list(res["pplot"]),
            list(res["svmplot"]), list(res["triaxplot"]), list(res["ecrplot"]), list(res["csrplot"]), res["fail"],
            res["nocoord_old"])


_model_cache = {}


def _digest(a, dtype):
    import hashlib
    return hashlib.blake2b(np.ascontiguousarray(a, dtype=dtype).reshape(-1).view(np.uint8), digest_size=16).digest()


def _model_for(elNodes, nocoord, materialbyElement, fix=None):
    """GPU model for the host-array mirrors below, cached on CONTENT: connectivity, the whole material row and the
    supports select the model; coordinates are compared on every hit and re-uploaded when they changed (calcDisp
    itself adds the imperfection to nocoord in place, as the reference does, fcVM.py:1240)."""
    mat = tuple(float(v) for v in np.asarray(materialbyElement, dtype=np.float64)[0][:3])
    key = (_digest(elNodes, np.int64), mat, tuple(sorted((int(k), float(v)) for k, v in (fix or {}).items())))
    xyz = np.ascontiguousarray(nocoord, dtype=np.float64)
    dx = _digest(xyz, np.float64)
    m = _model_cache.get(key)
    if m is None:
        _model_cache.clear()
        m = GpuModel(elNodes, xyz, mat, fix or {}, device=settings["device"])
        m._el_digest, m._xyz_digest = key[0], dx
        _model_cache[key] = m
    elif m._xyz_digest != dx:
        m.ctx.set_coords(xyz)
        m.nocoord = xyz.copy()
        m._ext = None
        m._xyz_digest = dx
    return m


def update_stress_load(gp10, elNodes, nocoord, materialbyElement, sig_yield, disp_new, du, sig, sig_update,
                       sig_test_global, qin, Et_E, LD, pgp):
    """fcVM.py:2197-2464 with host arrays: sig_update, sig_test_global, pgp are overwritten, the
    internal forces are ADDED to qin (as the reference does)."""
    c = _model_for(elNodes, nocoord, materialbyElement).ctx
    d = lambda a: c.array(a)  # noqa: E731
    s_new, s_test, q = c.empty(len(sig)), c.empty(len(sig)), c.empty(len(qin))
    pg = c.empty(len(pgp), np.uint8)
    c.update_stress_load(d(sig_yield), d(disp_new), d(du), d(sig), s_new, s_test, q, Et_E, LD, pg)
    sig_update[:] = s_new.get()
    sig_test_global[:] = s_test.get()
    qin += q.get()
    pgp[:] = pg.get().astype(pgp.dtype)


def update_PEEQ_CSR(nelem, materialbyElement, sig_test, sig_new, sig_yield, ultimate_strain, peeq, csr, triax,
                    pressure, sigmises, ecr, Et_E, elNodes=None, nocoord=None):
    """fcVM.py:2085-2137 with host arrays (updated in place).  Needs a context: pass the mesh of
    the last model or call after calcGSM."""
    if not _model_cache and (elNodes is None or nocoord is None):
        raise FcvmError("update_PEEQ_CSR: no GPU model yet (call calcGSM / update_stress_load first)")
    m = next(iter(_model_cache.values())) if elNodes is None else _model_for(elNodes, nocoord, materialbyElement)
    c = m.ctx
    dev = {k: c.array(v) for k, v in dict(sy=sig_yield, peeq=peeq, csr=csr, triax=triax, pressure=pressure,
                                           sigmises=sigmises, ecr=ecr).items()}
    c.update_peeq_csr(c.array(sig_test), c.array(sig_new), dev["sy"], ultimate_strain, dev["peeq"], dev["csr"],
                      dev["triax"], dev["pressure"], dev["sigmises"], dev["ecr"], Et_E)
    sig_yield[:] = dev["sy"].get(); peeq[:] = dev["peeq"].get(); csr[:] = dev["csr"].get()
    triax[:] = dev["triax"].get(); pressure[:] = dev["pressure"].get(); sigmises[:] = dev["sigmises"].get()
    ecr[:] = dev["ecr"].get()


def mapStresses(averaged, elNodes, nocoord, sig, peeq, sigvm, csr, noce, sig_yield, materialbyElement=((1.0, 0.0, 0.0),)):
    """fcVM.py:2497-2554.  Returns (tet10stress [nn,6], tet10peeq, tet10csr, tet10svm, tet10triax)."""
    de = _digest(elNodes, np.int64)
    for m in _model_cache.values():   # node -> element incidence is all this needs: any cached model of the same connectivity
        if m._el_digest == de and m.nn == len(nocoord):
            break
    else:
        m = _model_for(elNodes, nocoord, materialbyElement)
    c = m.ctx
    o = c.map_stresses(averaged, c.array(sig), c.array(peeq), c.array(sigvm), c.array(csr), sig_yield)
    return o[0].get().reshape(-1, 6), o[1].get(), o[2].get(), o[3].get(), o[4].get()


def calculate_principal_stress(tet10stress):
    """fcVM.py:2954-2994: (tet10s1, tet10s2, tet10s3, s1, s2, s3)."""
    if not _model_cache:
        raise FcvmError("calculate_principal_stress: no GPU context yet")
    c = next(iter(_model_cache.values())).ctx
    ps, pv = c.principal_stress(c.array(np.asarray(tet10stress).reshape(-1)))
    ps, pv = ps.get().reshape(-1, 3), pv.get().reshape(-1, 3, 3)
    return ps[:, 0].copy(), ps[:, 1].copy(), ps[:, 2].copy(), pv[:, 0].copy(), pv[:, 1].copy(), pv[:, 2].copy()


def exportVTK(gnl, elNodes, nocoord, nocoord_old, dis, disp_el, eigenval, eigenvec, tet10stress, tet10peeq, tet10csr,
              tet10svm, tet10triax, fy, file, maxImp, nstep, principal=None):
    """fcVM.py:2903-2950: same point-data fields, same order, legacy VTK 5.1 binary."""
    nn = len(nocoord)
    if principal is None:
        principal = calculate_principal_stress(tet10stress)
    s1, s2, s3, v1, v2, v3 = principal
    data = {}
    data["Critical Strain Ratio\n"] = np.asarray(tet10csr).reshape(-1)
    data["Equivalent Plastic Strain\n"] = np.asarray(tet10peeq).reshape(-1)
    data["von Mises Stress\n"] = np.asarray(tet10svm).reshape(-1)
    data["Triaxiality\n"] = np.asarray(tet10triax).reshape(-1)
    data["Displacement"] = np.asarray(dis).reshape(nn, 3)
    if gnl == "GNLY" and not (float(nstep) > 1.0 and float(maxImp) == 0.0):  # fcVM.py:2924-2931
        ev = np.asarray(eigenvec)
        data["Elastic Displacement"] = np.asarray(disp_el).reshape(nn, 3)
        for k in range(2):
            shape = ev[:, k] / np.max(np.abs(ev[:, k]))
            data["Buckling shape for lambda%d = %s" % (k + 1, str(round(float(eigenval[k]), 3)))] = shape.reshape(nn, 3)
    data["Stress Tensor"] = np.asarray(tet10stress).reshape(nn, 6)
    data["Major Principal Stress\n"] = s1
    data["Intermediate Principal Stress\n"] = s2
    data["Minor Principal Stress\n"] = s3
    data["Major Principal Stress Vector"] = v1
    data["Intermediate Principal Stress Vector"] = v2
    data["Minor Principal Stress Vector"] = v3
    vtkio.write_vtk(file, nocoord, np.asarray(elNodes) - 1, data)


def run(name, elNodes, nocoord, materialbyElement, fix, loads=None, control_dir=".", output_dir=".", ctl=None,
        window=None):
    """Headless equivalent of fcVM.FCMacro:70-243 for a mesh that is already in memory: reads
    <control_dir>/<name>.inp, solves on the GPU, writes <output_dir>/<name>.out and .vtk."""
    from fcvm_hip.solver import run_analysis
    if ctl is None:
        ctl = inpout.read_inp(os.path.join(control_dir, name + ".inp"))
    res = run_analysis(elNodes, nocoord, np.asarray(materialbyElement)[0], fix, loads or Loads(), ctl,
                       device=settings["device"], log=print, rtol=settings["rtol"])
    _model_cache.clear()
    res["model"]._el_digest = _digest(res["model"].elNodes, np.int64)
    res["model"]._xyz_digest = _digest(res["model"].nocoord, np.float64)
    _model_cache["run"] = res["model"]
    inpout.write_out(os.path.join(output_dir, name + ".out"), name, len(elNodes), len(nocoord), ctl["gnl"],
                     ctl["nstep"], res["eigenval"], res["loadsum"], res["crip"], res["x"], res["lout"], res["un"],
                     res["peeqplot"], res["pplot"], res["svmplot"], res["triaxplot"], res["ecrplot"], res["csrplot"])
    pr = res["principal"], res["principal_vec"]
    exportVTK(ctl["gnl"], elNodes, res["nocoord"], res["nocoord_old"], res["displacements"], res["disp_el"], res["eigenval"],
              res["eigenvec"],
              res["tet10stress"], res["tet10peeq"], res["tet10csr"], res["tet10svm"], res["tet10triax"],
              ctl["sig_yield"], os.path.join(output_dir, name + ".vtk"), ctl["maxImp"], ctl["nstep"],
              principal=(pr[0][:, 0], pr[0][:, 1], pr[0][:, 2], pr[1][:, 0], pr[1][:, 1], pr[1][:, 2]))
    return res


def run_fcstd(fcstd_path, control_dir, output_dir, name=None):
    """FreeCAD documents end to end without FreeCAD: <name>.FCStd (embedded mesh, BREP geometry and constraints, read
    by fcvm_hip.fcstd / fcvm_hip.brep: shapes bounded by planes and cylinders) + <control_dir>/<name>.inp ->
    <output_dir>/<name>.out / .vtk.  Replaces setUpAnalysis + setUpInput (fcVM.py:87-347) in front of the solver
    path.  The model name written into the .out header is the document's label, as in the macro."""
    from fcvm_hip import fcstd
    m = fcstd.load_model(fcstd_path)
    name = name or os.path.splitext(os.path.basename(fcstd_path))[0]
    return run(name, m["elNodes"], m["nocoord"], [m["material"]], m["fix"], m["loads"], control_dir, output_dir)


def calcSum(Edge_Elements, Face_Elements, mesh, CSR, peeq, svm):
    """fcVM.py:2830-2900: length / area weighted averages of the nodal results over selected CAD
    edges (lists of 3-node line elements) and faces (lists of 6-node triangles).  `mesh` is either
    the nodal coordinate array [nn,3] (node numbers 1-based as everywhere) or an object with a
    `.Nodes` mapping node -> vector with x, y, z (FreeCAD's FemMesh).  Host-side post-processing
    of O(selection) size."""
    from fcvm_hip.loads import GP2, GP6, _tri6
    if hasattr(mesh, "Nodes"):
        nodes = mesh.Nodes
        xyz = lambda n: np.array([nodes[n].x, nodes[n].y, nodes[n].z])  # noqa: E731
    else:
        arr = np.asarray(mesh, dtype=np.float64)
        xyz = lambda n: arr[n - 1]  # noqa: E731
    CSR, peeq, svm = np.asarray(CSR, float), np.asarray(peeq, float), np.asarray(svm, float)

    def integrate(groups, nper, weights):
        size, a_peeq, a_csr, a_svm = [], [], [], []
        for group in groups:
            tot = np.zeros(4)
            for element in group:
                x = np.array([xyz(n) for n in element])            # [nper, 3]
                nd = np.asarray(element) - 1
                for shp, jac, w in weights(x):
                    d = shp * abs(jac) * w
                    tot += np.array([d.sum(), (peeq[nd] * d).sum(), (CSR[nd] * d).sum(), (svm[nd] * d).sum()])
            if tot[0] > 0.0:
                tot[1:] /= tot[0]
            size.append(tot[0]); a_peeq.append(tot[1]); a_csr.append(tot[2]); a_svm.append(tot[3])
        return size, a_peeq, a_csr, a_svm

    def line_weights(x):
        for xi, w in GP2:  # shape2lin, fcVM.py:545-565
            shp = np.array([-0.5 * (1.0 - xi) * xi, 0.5 * (1.0 + xi) * xi, (1.0 + xi) * (1.0 - xi)])
            d = np.array([xi - 0.5, xi + 0.5, -2.0 * xi])
            yield shp, np.linalg.norm(d @ x), w

    def face_weights(x):
        for xi, et, w in GP6:  # shape6tri, fcVM.py:485-541
            shp, d = _tri6(xi, et)
            yield shp, np.linalg.norm(np.cross(d[0] @ x, d[1] @ x)), w

    el, ep, ec, es = integrate(Edge_Elements, 3, line_weights)
    fa, fp, fc, fs = integrate(Face_Elements, 6, face_weights)
    return el, ep, ec, es, fa, fp, fc, fs


def write_avr(path, name, Edge_Names, Face_Names, edge_length, edge_peeq, edge_CSR, edge_svm, face_area, face_peeq,
              face_CSR, face_svm):
    """The .avr summary written by fcVM_sum.FCMacro:110-160."""
    sep = "\n=========================================================================================================================\n\n"
    with open(path, "w") as f:
        f.write("model name:{0: >30}\n\n".format(name))
        f.write("average values")
        f.write(sep)
        f.write("{0: >10}{1: >10}{2: >10}{3: >10}    {4}\n".format("Length", "peeq", "CSR", "svmises", "edge"))
        for i, edge in enumerate(Edge_Names):
            f.write("{0: >10.2e}{1: >10.2e}{2: >10.2e}{3: >10.2e}    {4}\n".format(edge_length[i], edge_peeq[i], edge_CSR[i],
                                                                             edge_svm[i], edge))
        f.write(sep)
        f.write("{0: >10}{1: >10}{2: >10}{3: >10}    {4}\n".format("Area", "peeq", "CSR", "svmises", "face"))
        for i, face in enumerate(Face_Names):
            f.write("{0: >10.2e}{1: >10.2e}{2: >10.2e}{3: >10.2e}    {4}\n".format(face_area[i], face_peeq[i], face_CSR[i],
                                                                             face_svm[i], face))
        f.write(sep)
