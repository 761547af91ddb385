#!/usr/bin/env python3
"""Shader / memory clock, power and temperature of the visible GPUs while a workload runs (sysfs hwmon files, no tools):
is the SpMV's 0.52 <-> 0.62 ms a clock state?

  python3 tools/clock_log.py --what spmv --seconds 12      # back-to-back SpMV launches, sampled every 50 ms
  python3 tools/clock_log.py --what pcg --seconds 12       # Newton iterations of the bench workload
"""
import argparse
import glob
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
sys.path.insert(0, REPO)


def sensors():
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        dev = os.path.join(card, "device")
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            files = {}
            for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input"):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    files[name] = p
            if files:
                out.append((os.path.basename(card), files))
    return out


def read(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except Exception:
        return float("nan")


def dpm(card):
    res = {}
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        p = "/sys/class/drm/%s/device/%s" % (card, name)
        try:
            cur = [ln for ln in open(p).read().splitlines() if ln.strip().endswith("*")]
            res[name] = cur[0] if cur else "?"
        except Exception as e:
            res[name] = "unreadable (%s)" % type(e).__name__
    return res


ap = argparse.ArgumentParser()
ap.add_argument("--what", default="spmv")
ap.add_argument("--seconds", type=float, default=12.0)
ap.add_argument("--cells", type=int, default=55)
ap.add_argument("--period", type=float, default=0.05)
a = ap.parse_args()

sens = sensors()
print("sensors:", [(c, sorted(f)) for c, f in sens])
for c, _ in sens:
    print(c, dpm(c))
samples, marks, stop = [], [], False


def poll():
    while not stop:
        t = time.perf_counter()
        row = [t]
        for c, f in sens:
            row += [read(f.get("freq1_input", "")) / 1e6, read(f.get("freq2_input", "")) / 1e6,
                    read(f.get("power1_average", f.get("power1_input", ""))) / 1e6, read(f.get("temp1_input", "")) / 1e3]
        samples.append(row)
        time.sleep(a.period)


from fcvm_hip import meshgen  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel, RiksDriver  # noqa: E402

case = meshgen.cube_case(a.cells, l=100.0, traction=(20.0, 0.0, 200.0))
m = GpuModel(case["elNodes"], case["nocoord"], (210000.0, 0.3, 7.8e-6), case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
th = threading.Thread(target=poll, daemon=True)
if a.what == "spmv":
    c.set_precond("bj")
    m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    x, y = c.array(np.random.default_rng(0).standard_normal(m.ndof)), c.empty(m.ndof)
    c.spmv(x, y); c.sync()
    time.sleep(3.0)                       # idle: let the clocks settle
    th.start()
    time.sleep(0.5)
    t_end = time.perf_counter() + a.seconds
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        for _ in range(50):
            c.spmv(x, y)
        c.sync()
        t1 = time.perf_counter()
        marks.append((t0, t1, (t1 - t0) / 50 * 1e3))
else:
    import bench
    stats = {}
    drv = RiksDriver(m, bench.CTL, rtol=bench.RTOL, stats=stats)
    drv.start_step()
    c.sync()
    time.sleep(3.0)
    th.start()
    time.sleep(0.5)
    t_end = time.perf_counter() + a.seconds
    while time.perf_counter() < t_end:
        s0 = dict(stats)
        t0 = time.perf_counter()
        drv.iteration()
        c.sync()
        t1 = time.perf_counter()
        marks.append((t0, t1, (stats["spmv_ms"] - s0["spmv_ms"]) / max(1, stats["spmv_launches"] - s0["spmv_launches"])))
time.sleep(0.5)
stop = True
th.join()
S = np.array(samples)
ncard = len(sens)
# the card that works is the one whose power moves
busy = int(np.nanargmax([np.nanmax(S[:, 1 + 4 * k + 2]) - np.nanmin(S[:, 1 + 4 * k + 2]) for k in range(ncard)])) if ncard else -1
print("busiest card:", sens[busy][0] if ncard else None, dpm(sens[busy][0]) if ncard else None)
print("%8s %10s %10s %9s %7s   %s" % ("t [s]", "freq1 MHz", "freq2 MHz", "power W", "temp C", "SpMV ms (window ending here)"))
t00 = S[0, 0]
mi = 0
for row in S[:: max(1, int(0.25 / a.period))]:
    while mi + 1 < len(marks) and marks[mi][1] < row[0]:
        mi += 1
    ms = marks[mi][2] if marks and marks[mi][0] <= row[0] else float("nan")
    k = busy
    print("%8.2f %10.0f %10.0f %9.1f %7.1f   %.4f" % (row[0] - t00, row[1 + 4 * k], row[2 + 4 * k], row[3 + 4 * k], row[4 + 4 * k], ms))
