"""Round 5, verdict item 3: the FP64 SpMV of the symmetric window form is 0.335 ms in some processes and 0.38 ms in others ON THE SAME BOX with
the same binary (tools/sessions/gpu_r05d.sh), from cold, at any power (gpu_r05f.sh), whatever the instruction mix (A/B builds): the placement of
a process's allocations.  Does the order in which the workgroups walk the matrix -- eight streams an eighth of the value array apart (mode 0)
against one moving front (identity / chunks per XCD) -- take the placement out of it?  Several model instances in ONE process (each allocates
its own 2.1 GB value array), every map mode timed on each: FP64 SpMV and 16-bit pass back to back, and the real in-PCG figure."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples  # noqa: E402
from fcvm_hip._lib import call  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

MODES = [int(v) for v in os.environ.get("GMAP_MODES", "0,1,4,16,64").split(",")]
NINST = int(os.environ.get("GMAP_INSTANCES", "4"))
case = examples.plate_with_hole(1.45)
keep = []
for inst in range(NINST):
    m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    keep.append(m)          # (instances stay alive: the next one's arrays land somewhere else)
    c = m.ctx
    glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    x, y, z = c.array(np.random.default_rng(0).standard_normal(m.ndof)), c.empty(m.ndof), c.empty(m.ndof)
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    for mode in MODES:
        call("fcvm_set_group_map", c.h, mode)
        for _ in range(5):
            c.spmv(x, y)
        c.sync(); t0 = time.perf_counter()
        for _ in range(50):
            c.spmv(x, y)
        c.sync(); t_sp = (time.perf_counter() - t0) / 50 * 1e3
        t0 = time.perf_counter()
        for _ in range(50):
            c.spmv_lo(x, z)
        c.sync(); t_lo = (time.perf_counter() - t0) / 50 * 1e3
        u, info = c.solve(f, rtol=1e-10)
        u, info = c.solve(f, rtol=1e-10)
        tm = c.solve_timing()
        print("instance %d  map %3d   FP64 SpMV back to back %.4f ms   16-bit pass %.4f ms   in PCG %.4f ms (%d iterations, solve %.1f ms)" %
              (inst, mode, t_sp, t_lo, tm["spmv_ms"] / max(tm["spmv_timed"], 1), info["iterations"], tm["total_ms"]), flush=True)
    call("fcvm_set_group_map", c.h, 0)
print("stream read %.0f GB/s" % keep[0].ctx.membench(3 << 30))
