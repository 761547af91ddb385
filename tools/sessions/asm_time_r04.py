"""assembly time on the 1 M plate: straight into the symmetric window form against the sliced-ELL + copy route (FCVM_ASSEMBLE_SYM=0)
   python tools/sessions/asm_time_r04.py [tangent]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.device import Context
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
tangent = len(sys.argv) > 1 and sys.argv[1] == "tangent"
h = float(os.environ.get("PLATE_H", "1.45"))
case = examples.plate_with_hole(h)
ne, nn = len(case["elNodes"]), len(case["nocoord"])
rng = np.random.default_rng(1)
for route in os.environ.get("ROUTES", "direct,copy").split(","):
    os.environ["FCVM_ASSEMBLE_SYM"] = "1" if route == "direct" else "0"
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]), ctx=Context(0))
    c = m.ctx
    kw = {}
    if tangent:
        kw = dict(disp=c.array(rng.standard_normal(3 * nn) * 1e-3), sig=c.array(rng.standard_normal(24 * ne) * 50.0),
                  pgp=c.array((rng.random(4 * ne) < 0.3).astype(np.uint8), np.uint8), Et_E=0.0)
    ts = []
    for rep in range(6):
        c.sync(); t0 = time.perf_counter()
        m.calc_gsm((0.0, 0.0, 0.0), want_x=False, **kw)
        c.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s %s: calc_gsm %s ms (first call incl. hierarchy set-up); %s" % (route, "tangent" if tangent else "elastic",
          " ".join("%.2f" % t for t in ts), {k: v for k, v in c.spmv_sym_info().items() if k in ("assembled_directly", "blocks")}), flush=True)
    c.close()
