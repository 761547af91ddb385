#!/usr/bin/env python3
"""SpMV rate on the (refined) Embankment mesh -- unstructured node numbering."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples
from fcvm_hip.loads import Loads
from fcvm_hip.solver import GpuModel
refine = int(sys.argv[1]) if len(sys.argv) > 1 else 3
case = examples.embankment(refine)
el, xyz = case["elNodes"], case["nocoord"]
if len(sys.argv) > 2 and sys.argv[2] == "morton":
    q = ((xyz - xyz.min(0)) / (xyz.max(0) - xyz.min(0)) * 1023).astype(np.uint64)
    def spread(v):
        v = (v | (v << 32)) & 0x1f00000000ffff
        v = (v | (v << 16)) & 0x1f0000ff0000ff
        v = (v | (v << 8)) & 0x100f00f00f00f00f
        v = (v | (v << 4)) & 0x10c30c30c30c30c3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    order = np.argsort(key, kind="stable")           # new position -> old node
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    xyz = xyz[order]; el = inv[el - 1] + 1
    fix = {3 * int(inv[d // 3]) + d % 3: v for d, v in case["fix"].items()}
else:
    fix = case["fix"]
m = GpuModel(el, xyz, case["material"], fix, Loads())
c = m.ctx
c.set_precond("bj")
m.calc_gsm((0.0, 0.0, -10.0), want_x=False)
info = c.pattern_info()
x = c.array(np.random.default_rng(0).standard_normal(m.ndof)); y = c.empty(m.ndof)
for _ in range(5): c.spmv(x, y)
c.sync(); t0 = time.perf_counter()
for _ in range(200): c.spmv(x, y)
c.sync(); ms = (time.perf_counter() - t0) / 200 * 1e3
alg = info["nnz_blocks"] * 76 + 2 * info["ndof"] * 8
print("elements %d nnzb %d padded %d (%.1f%% padding) spmv %.4f ms  %.0f GB/s algorithmic" % (m.ne, info["nnz_blocks"], info["padded_slots"], 100.0 * (info["padded_slots"] / info["nnz_blocks"] - 1), ms, alg / ms / 1e6))
