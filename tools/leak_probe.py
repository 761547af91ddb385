#!/usr/bin/env python3
"""Create, use and close many contexts in one process: does device memory come back?  (rocm-smi VRAM before / after)"""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import meshgen  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402


def vram():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
    for ln in out.splitlines():
        if "Used" in ln:
            return int(ln.split(":")[-1]) / 1e9
    return float("nan")


case = meshgen.cube_case(24, l=100.0, traction=(20.0, 0.0, 200.0))
print("before: %.2f GB" % vram())
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    m = GpuModel(case["elNodes"], case["nocoord"], (210000.0, 0.3, 7.8e-6), case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    f = m.ctx.mul(m.fixdof, glv)
    u, info = m.ctx.solve(f, rtol=1e-8)
    m.ctx.close()
    del m, glv, modf, f, u
    if k % 10 == 9:
        print("after %d contexts: %.2f GB (last solve %d iterations)" % (k + 1, vram(), info["iterations"]))
