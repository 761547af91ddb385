"""Round 5, verdict item 3 (last step): is the "slow box" a thermal / power state of the memory system?  One process, 1 M plate, product library:
back-to-back time of the FP64 SpMV (50 launches), of the 16-bit pass and of the plain stream read, sampled (1) at the start, (2) after each of six
5-second bursts of full PCG load (elastic solves one after the other: ~1.1 kW), (3) after 10 / 20 / 30 s of idling -- with every temperature,
clock and power sensor the card's hwmon directory offers."""
import glob
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402


def rd(p):
    try:
        return open(p).read().strip()
    except Exception:
        return "?"


hw = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
print("hwmon dirs:", hw)
for h in hw:
    print(h, {os.path.basename(p): rd(p) for p in sorted(glob.glob(h + "/*_label"))})


def sensors():
    out = []
    for h in hw:
        row = {}
        for p in sorted(glob.glob(h + "/temp*_input")) + sorted(glob.glob(h + "/freq*_input")) + sorted(glob.glob(h + "/power*_average")) + sorted(glob.glob(h + "/power*_input")):
            v = rd(p)
            try:
                v = float(v)
                v = v / 1e3 if "temp" in p else v / 1e6
            except ValueError:
                pass
            row[os.path.basename(p).replace("_input", "").replace("_average", "")] = v
        out.append(row)
    return out


case = examples.plate_with_hole(1.45)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
x, y, z = c.array(np.random.default_rng(0).standard_normal(m.ndof)), c.empty(m.ndof), c.empty(m.ndof)
f = c.mul(m.fixdof, glv)
c.axpby(1.0, f, 1.0, modf, out=f)
t_start = time.perf_counter()


def probe(tag):
    for _ in range(3):
        c.spmv(x, y)
    c.sync(); t0 = time.perf_counter()
    for _ in range(50):
        c.spmv(x, y)
    c.sync(); t_sp = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(50):
        c.spmv_lo(x, z)
    c.sync(); t_lo = (time.perf_counter() - t0) / 50 * 1e3
    s = sensors()
    bw = c.membench(3 << 30)
    busy = max(s, key=lambda r: max([v for k, v in r.items() if k.startswith("power") and isinstance(v, float)] or [0.0])) if s else {}
    print("%6.1f s  %-28s FP64 SpMV %.4f ms  16-bit pass %.4f ms  stream %.0f GB/s  | %s" %
          (time.perf_counter() - t_start, tag, t_sp, t_lo, bw, " ".join("%s %.0f" % (k, v) for k, v in busy.items() if isinstance(v, float))), flush=True)


probe("start")
for k in range(6):
    t0 = time.perf_counter()
    n, ms = 0, 0.0
    while time.perf_counter() - t0 < 5.0:
        u, info = c.solve(f, rtol=1e-10)
        tm = c.solve_timing()
        ms += tm["spmv_ms"]; n += tm["spmv_timed"]
    probe("after burst %d (in PCG %.4f)" % (k + 1, ms / max(n, 1)))
for k in range(3):
    time.sleep(10.0)
    probe("after %d s idle" % (10 * (k + 1)))
