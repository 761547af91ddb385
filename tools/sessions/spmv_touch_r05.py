"""Round 5, verdict item 3: inside PCG the FP64 SpMV runs at ~0.385 ms even in processes where it takes 0.335 ms back to back.  Between two of
its launches a PCG iteration streams 1.5 GB of other arrays.  Do cold address-translation caches cost the difference?  FCVM_SPMV_TOUCH_KB=k makes
the PCG touch one word per k KB of the value / column arrays (from 16 workgroups = all XCDs) right before each SpMV.  Per model instance: back to
back time, then solves with touch off / 2048 / 64 / 4 KB / off again: in-PCG SpMV time (HIP events) and the whole solve."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

case = examples.plate_with_hole(1.45)
keep = []
for inst in range(int(os.environ.get("TOUCH_INSTANCES", "3"))):
    os.environ.pop("FCVM_SPMV_TOUCH_KB", None)
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    keep.append(m)
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    x, y = c.array(np.random.default_rng(0).standard_normal(m.ndof)), c.empty(m.ndof)
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    for _ in range(5):
        c.spmv(x, y)
    c.sync(); t0 = time.perf_counter()
    for _ in range(50):
        c.spmv(x, y)
    c.sync()
    print("instance %d  back to back %.4f ms" % (inst, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
    for kb in ("0", "2048", "64", "4", "0"):
        os.environ["FCVM_SPMV_TOUCH_KB"] = kb
        for _ in range(3):
            u, info = c.solve(f, rtol=1e-10)
        tm = c.solve_timing()
        print("instance %d  touch every %5s KB: in-PCG SpMV %.4f ms, solve %.2f ms (%d iterations)" %
              (inst, kb, tm["spmv_ms"] / max(tm["spmv_timed"], 1), tm["total_ms"], info["iterations"]), flush=True)
