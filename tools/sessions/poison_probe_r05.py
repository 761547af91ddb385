"""FCVM_POISON=1 probe: which operation of the default path turns NaN on a mesh when fresh allocations (and 512 bytes behind floating-point arrays)
are filled with 0xFF?  usage: poison_probe_r05.py emb:1 | plate:8 | cube:12"""
import os, sys
import numpy as np
os.environ["FCVM_POISON"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples, meshgen
from fcvm_hip.device import Context
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
kind, arg = sys.argv[1].split(":")
if kind == "emb":
    case = examples.embankment(refine=int(arg)); ld = Loads(); grav = (0.0, 0.0, -10.0)
elif kind == "plate":
    case = examples.plate_with_hole(float(arg)); ld = Loads(ufaces=case["ufaces"], uloads=case["uloads"]); grav = (0.0, 0.0, 0.0)
else:      # the cube of configs[4] at n^3 cells: base clamped, top face moved down
    n = int(arg)
    el, xyz = meshgen.cube_tet10(n, n, n, 100.0, 100.0, 100.0)
    fix = {3 * int(nd) + k: 0.0 for nd in meshgen.nodes_on_plane(xyz, 2, 0.0) for k in range(3)}
    for nd in meshgen.nodes_on_plane(xyz, 2, 100.0):
        fix[3 * int(nd) + 2] = -0.08
    case = dict(elNodes=el, nocoord=xyz, material=(210000.0, 0.3, 7.8e-6), fix=fix); ld = Loads(); grav = (0.0, 0.0, 0.0)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], ld, ctx=Context(0))
c = m.ctx
def bad(name, v):
    a = v.get()
    nb = int((~np.isfinite(a)).sum())
    print("%-28s %s" % (name, "finite" if nb == 0 else "%d of %d NOT finite, first at %s" % (nb, a.size, np.where(~np.isfinite(a))[0][:6])), flush=True)
    return nb
glv, modf, _, _ = m.calc_gsm(grav, want_x=False)
bad("glv", glv); bad("modf", modf)
bad("fixdof", m.fixdof)
t_ = c.mul(m.fixdof, glv)
bad("fixdof * glv", t_)
f = c.axpby(1.0, t_, 1.0, modf)
bad("f", f)
bad("glv again", glv); bad("modf again", modf)
print(c.precond_info())
bad("K f", c.spmv(f, c.empty(m.ndof)))
bad("K~ f", c.spmv_lo(f, c.empty(m.ndof)))
bad("diagonal blocks", c.get_diag_blocks())
bad("block-Jacobi f", c.precond(f, c.empty(m.ndof)))
bad("M^-1 f", c.mg_apply(f, c.empty(m.ndof)))
try:
    u, info = c.solve(f, rtol=1e-10)
    bad("u", u); print(info)
except Exception as e:
    print("solve:", str(e)[-200:])
if len(sys.argv) > 2 and sys.argv[2] == "analysis":
    from fcvm_hip import solver, _lib
    c.close()
    # every library call followed by a check of the vectors the driver holds: the first call after which one of them is not finite
    names = ["ue", "du", "disp_new", "disp_old", "sig_old", "sig_new", "sig_test", "qin", "r", "tmp", "peeq", "csr", "triax", "pressure", "sigmises", "ecr", "sig_yield", "glv", "modf", "f", "a"]
    orig = _lib.call
    state = {"drv": None, "hit": False}
    def traced(name, *a):
        rc = orig(name, *a)
        d = state["drv"]
        if d is not None and not state["hit"] and not name.startswith(("fcvm_d2h", "fcvm_h2d", "fcvm_malloc", "fcvm_free", "fcvm_memset")):
            for nm in names:
                v = getattr(d, nm, None)
                if v is not None and hasattr(v, "get"):
                    arr = v.get()
                    if not np.isfinite(arr).all():
                        print("after %s: %s has %d non-finite entries of %d (first %s)" % (name, nm, (~np.isfinite(arr)).sum(), arr.size, np.where(~np.isfinite(arr))[0][:5]), flush=True)
                        state["hit"] = True
        return rc
    import fcvm_hip.device as dev
    dev.call = traced; solver.call = traced if hasattr(solver, "call") else None
    Drv = solver.RiksDriver
    print("driver class:", Drv)
    if Drv is not None:
        oi = Drv.__init__
        def init(self, *a, **k):
            state["drv"] = self; oi(self, *a, **k)
        Drv.__init__ = init
    res = solver.run_analysis(case["elNodes"], case["nocoord"], case["material"], case["fix"], ld, dict(case["ctl"], nstep=4), rtol=1e-11)
    print({k: (bool(np.isfinite(np.asarray(v, float)).all()) if isinstance(v, (np.ndarray, list)) else v) for k, v in res.items() if k in ("lout", "un", "displacements", "peeq", "crip")})
