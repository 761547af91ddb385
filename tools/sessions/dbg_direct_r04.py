"""debug: assembly straight into the symmetric window form against the copy route (block-Jacobi context: no hierarchy set-up)"""
import os, sys
import numpy as np
import scipy.sparse as sp
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ["FCVM_PRECOND"] = "bj"
from conftest import small_case
from fcvm_hip.device import Context
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
case = small_case(7)
n = 3 * len(case["nocoord"])
out = {}
for route in ("copy", "direct"):
    os.environ["FCVM_ASSEMBLE_SYM"] = "1" if route == "direct" else "0"
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctx=Context(0))
    c = m.ctx
    glv, modf, V, xgp = m.calc_gsm((0.3, -0.2, -9.81))
    print(route, c.spmv_sym_info())
    r, cc, v = c.matrix_coo()
    K = sp.coo_matrix((v, (r, cc)), shape=(n, n)).tocsr()
    x = c.array(m.fixdof_h * np.random.default_rng(1).standard_normal(n))
    out[route] = dict(K=K, dblk=c.get_diag_blocks().get(), y=c.spmv(x, c.empty(n)).get(), glv=glv.get(), modf=modf.get(), V=V)
    for k, a in out[route].items():
        if k != "K" and k != "V":
            print("  ", k, "nan:", int(np.isnan(a).sum()), "absmax %.4g" % np.nanmax(np.abs(a)))
    print("   K nan:", int(np.isnan(K.data).sum()), "absmax %.4g" % np.nanmax(np.abs(K.data)), "V", V)
    c.close()
d, cp = out["direct"], out["copy"]
D = abs(d["K"] - cp["K"])
print("K diff max", D.max(), "of", abs(cp["K"]).max(), "nonzero diffs", (D > 0).nnz if hasattr(D > 0, "nnz") else None)
coo = D.tocoo()
bad = coo.data > 1e-9 * abs(cp["K"]).max()
print("bad entries", bad.sum(), "rows (nodes)", np.unique(coo.row[bad] // 3)[:20], "cols", np.unique(coo.col[bad] // 3)[:20])
for k in ("dblk", "y", "glv", "modf"):
    print(k, "max diff", np.nanmax(np.abs(d[k] - cp[k])), "equal", np.array_equal(d[k], cp[k]))
