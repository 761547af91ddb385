"""Round 5, verdict item 3: what limits the FP64 SpMV of the symmetric window form on the pool's "slow" boxes (0.38-0.39 ms inside PCG against
0.34-0.35 on others, while the stream read and the 16-bit passes are the same)?  One library build per process (FCVM_HIP_LIB: the product or
an A/B build of tools/sessions/build_ab_r05.sh), 1 M plate:
  * back to back: 4 x 50 launches of the FP64 SpMV;
  * interleaved with the two 16-bit smoother passes of a PCG iteration (what the solver does between two FP64 launches): time of
    N x [FP64, 16-bit, 16-bit] minus N x [16-bit, 16-bit];
  * product library only: the real in-PCG figure from HIP events (an elastic solve);
  * shader clock / power from sysfs while the loops run."""
import glob
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
from fcvm_hip import examples  # noqa: E402
from fcvm_hip.loads import Loads  # noqa: E402
from fcvm_hip.solver import GpuModel  # noqa: E402

tag = os.environ.get("AB_TAG", "product")
real = tag in ("product", "minwaves3")       # builds whose results are right


def sensors():
    out = []
    for hw in glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"):
        f = {n: os.path.join(hw, n) for n in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input") if os.path.exists(os.path.join(hw, n))}
        if f:
            out.append(f)
    return out


def rd(p):
    try:
        return float(open(p).read().strip())
    except Exception:
        return float("nan")


sens, samples, stop = sensors(), [], [False]


def poll():
    while not stop[0]:
        row = []
        for f in sens:
            row += [rd(f.get("freq1_input", "")) / 1e6, rd(f.get("freq2_input", "")) / 1e6, rd(f.get("power1_average", f.get("power1_input", ""))) / 1e6]
        samples.append(row)
        time.sleep(0.02)


case = examples.plate_with_hole(1.45)
m = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
c = m.ctx
glv, modf, _, _ = m.calc_gsm((0.0, 0.0, 0.0), want_x=False)
x, y, z = c.array(np.random.default_rng(0).standard_normal(m.ndof)), c.empty(m.ndof), c.empty(m.ndof)
si = c.spmv_sym_info()
print("[%s] symmetric form in use %s, blocks %d" % (tag, si["in_use"], si["blocks"]), flush=True)
for _ in range(5):
    c.spmv(x, y); c.spmv_lo(x, z)
c.sync()
th = threading.Thread(target=poll, daemon=True)
th.start()
time.sleep(0.3)
n0 = len(samples)
b2b = []
for rep in range(4):
    c.sync(); t0 = time.perf_counter()
    for _ in range(50):
        c.spmv(x, y)
    c.sync()
    b2b.append((time.perf_counter() - t0) / 50 * 1e3)
n1 = len(samples)
N = 60
inter = []
for rep in range(3):
    c.sync(); t0 = time.perf_counter()
    for _ in range(N):
        c.spmv_lo(x, z); c.spmv_lo(z, y)
    c.sync(); t_lo = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(N):
        c.spmv(x, y); c.spmv_lo(x, z); c.spmv_lo(z, y)
    c.sync(); t_all = time.perf_counter() - t0
    inter.append(((t_all - t_lo) / N * 1e3, t_lo / (2 * N) * 1e3))
n2 = len(samples)
pcg = None
if real:
    f = c.mul(m.fixdof, glv)
    c.axpby(1.0, f, 1.0, modf, out=f)
    for _ in range(2):
        u, info = c.solve(f, rtol=1e-10)
        tm = c.solve_timing()
        pcg = (tm["spmv_ms"] / max(tm["spmv_timed"], 1), info["iterations"], tm["total_ms"])
n3 = len(samples)
stop[0] = True
th.join()
S = np.array(samples) if samples else np.zeros((0, 3))


def stat(a, b):
    if b <= a or S.shape[1] < 3:
        return "n/a"
    k = int(np.nanargmax(np.nanmax(S[:, 2::3], axis=0))) if S.shape[1] >= 3 else 0      # the card whose power is largest
    seg = S[a:b, 3 * k:3 * k + 3]
    return "sclk %.0f-%.0f MHz (mean %.0f), mclk %.0f, power %.0f-%.0f W (mean %.0f)" % (np.nanmin(seg[:, 0]), np.nanmax(seg[:, 0]), np.nanmean(seg[:, 0]),
                                                                                          np.nanmean(seg[:, 1]), np.nanmin(seg[:, 2]), np.nanmax(seg[:, 2]), np.nanmean(seg[:, 2]))


print("[%s] back to back      %s ms   | %s" % (tag, " ".join("%.4f" % v for v in b2b), stat(n0, n1)))
print("[%s] between 16-bit    %s ms (16-bit pass itself %s ms)   | %s" % (tag, " ".join("%.4f" % v[0] for v in inter), " ".join("%.4f" % v[1] for v in inter), stat(n1, n2)))
if pcg:
    print("[%s] inside PCG        %.4f ms over %d iterations (solve %.1f ms)   | %s" % (tag, pcg[0], pcg[1], pcg[2], stat(n2, n3)))
print("[%s] stream read %.0f GB/s" % (tag, c.membench(3 << 30)), flush=True)
