#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): Newton iterations/s and elements*GP/s of the elastoplastic tet10 solver on the reference's
Plate_with_hole_Example at 1 M tet10, SpMV GB/s against the HBM roofline.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--plate-h H]

N = 1: the workload is the reference's Plate_with_hole_Example (geometry, supports, pressure load, von Mises material, its shipped
control file) on a mapped mesh of 1,006,434 tet10.  The whole collapse analysis runs the way calcDisp does (fcVM.py:1304-1559); a
"step" is one Newton / Riks iteration of it (fcVM.py:1346-1455): multigrid-preconditioned CG solve of K due = relax * r to 1e-10
(replacing the CHOLMOD back substitution), arc-length correction, update_stress_load over all elements, out-of-balance norm.  The
timed region is Newton iterations W+1 .. W+K of that analysis with whatever start_step / end_step calls fall between them.  The
one-time work (pattern, K assembly = calcGSM, hierarchy, elastic solve) happens before it, where the reference does it (it factorises
once, fcVM.py:1121), and is reported separately in the same JSON line, as are: `roofline` (the PCG's SpMV, HIP events on the solver's
stream), `cpu_baseline` (the oracle port on this host), `north_star_workload` (the whole analysis), `configs2_cube` (BASELINE
configs[2]: the synthetic 1 M cube) and `gnly` (the geometrically non-linear path: tangent re-assembly + pair of solves per iteration).

N > 1 (one process per GPU under torch.distributed.run; fcvm_hip/dist_bench.py): the same analysis on a mesh of N x 1 M tet10 (weak
scaling), elements cut into N ranges along a Morton curve, RCCL point-to-point sums at the shared nodes after every product and two
small all-reduces per PCG iteration.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)

CTL = dict(sig_yield=120.0, grav_x=0.0, grav_y=0.0, grav_z=0.0, nstep=2, iterat_max=10 ** 9, error_max=0.0, relax=1.2,
           scale_re=2.0, scale_up=1.2, scale_dn=1.2, disp_output="total", ultimate_strain=0.25, Et_E=0.0,
           target_LF=1.0e9, csr_option="PEEQ", averaged_option="unaveraged", gnl="GNLN", maxImp="0.0", ev1="1.0",
           ev2="0.0")
MATERIAL = (210000.0, 0.3, 7.8e-6)
TRACTION = (20.0, 0.0, 200.0)
RTOL = 1.0e-10
# HBM bytes per SpMV launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, gfx950
# correction applied): bench.py cannot collect counters itself, so the measured figure is kept in a file together with
# the layout it was measured for (cells, storage order, kernel variant, commit) and is only quoted when this run uses
# that layout; otherwise `traffic` is null.
SPMV_TRAFFIC_FILE = os.path.join(REPO, "profiles", "spmv_traffic.json")


def spmv_traffic(cells, variant):
    """cells: the workload key of the measurement (mesh name; cube edge cells in the entries of rounds 1-2)"""
    try:
        t = json.load(open(SPMV_TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, None
    for e in reversed(t.get("measurements", [])):      # (the latest measurement of a layout)
        if e.get("workload", e.get("cells")) == cells and e.get("variant") == variant:
            return float(e["traffic_bytes_per_launch"]), "%s (commit %s, %s)" % (e.get("source", SPMV_TRAFFIC_FILE), e.get("commit", "?"), e.get("method", ""))
    return None, None


def _cpu_newton_iterations(orc, el, xyz, mat, fix, loads, iters, sig_yield):
    """The oracle (C restatement + SuperLU in place of CHOLMOD) on this host: one Newton iteration = back substitution +
    update_stress_load, factorisation outside (the reference factorises once, fcVM.py:1121).  Returns
    (seconds per iteration, calcGSM seconds, factorisation seconds)."""
    ne, nn = len(el), len(xyz)
    t0 = time.perf_counter()
    stm, row, col, glv, modf, *_ = orc.calcGSM(el, xyz, mat, fix, (0, 0, 0), loads)
    t_asm = time.perf_counter() - t0
    t0 = time.perf_counter()
    factor = orc._Direct(stm, row, col, 3 * nn)
    t_fac = time.perf_counter() - t0
    fixdof = np.ones(3 * nn)
    fixdof[list(fix.keys())] = 0.0
    ue = factor(fixdof * glv + modf)
    du = 0.5 * ue
    sig_old, sig_new, sig_test = np.zeros(24 * ne), np.zeros(24 * ne), np.zeros(24 * ne)
    sy = np.full(4 * ne, sig_yield)
    pgp = np.zeros(4 * ne, np.uint8)
    disp = np.zeros(3 * nn)
    qin = np.zeros(3 * nn)
    orc.update_stress_load(el, xyz, mat, sy, disp, du, sig_old, sig_new, sig_test, qin, 0.0, False, pgp)
    r = fixdof * (0.5 * glv - qin)
    t0 = time.perf_counter()
    for _ in range(iters):
        due = factor(1.2 * r)
        du = du + due
        qin = np.zeros(3 * nn)
        orc.update_stress_load(el, xyz, mat, sy, disp, du, sig_old, sig_new, sig_test, qin, 0.0, False, pgp)
        r = fixdof * (0.5 * glv - qin)
        np.linalg.norm(r)
    return (time.perf_counter() - t0) / iters, t_asm, t_fac


def cpu_baseline(plate_h=6.0, iters=5):
    """cpu_baseline leg (rank 0, N = 1): the oracle port timed on this host's cores on a bounded sample of the headline workload --
    the reference's Plate_with_hole_Example (shipped control file) on a mesh of the same generator at plate_h (6.0: 16,830
    tet10, ~7 s of factorisation; 4.0: 48,600 = BASELINE configs[1], ~160 s; profiles/r03_cpu_gpu_table.json holds three
    sizes).  The 1 M-element mesh itself is out of reach of a sparse direct solver on one core (hours, > 100 GB of fill).  One
    Newton iteration = back substitution + update_stress_load, as in the reference's loop; the GPU side of `same_mesh` is
    the whole collapse analysis on the SAME mesh through the C-ABI.  SuperLU (symmetric mode) stands in for the reference's
    CHOLMOD (fcVM.py:1121-1135; sksparse is not installed): CHOLMOD's supernodal back substitution is probably faster than
    SuperLU's, so the CPU figure here is likely lower than the reference's own would be."""
    from fcvm_hip import examples
    from oracle import oracle as orc
    pc = examples.plate_with_hole(plate_h)
    ne, nn = len(pc["elNodes"]), len(pc["nocoord"])
    pdt, p_asm, p_fac = _cpu_newton_iterations(orc, pc["elNodes"], pc["nocoord"], pc["material"], pc["fix"],
                                               orc.Loads(ufaces=pc["ufaces"], uloads=pc["uloads"]), iters, float(pc["ctl"]["sig_yield"]))
    gpu = collapse_run(pc, 0, None)["analysis"]
    return {"value": ne * 4 / pdt, "unit": "elements*GP/s", "cores": 1, "kind": "port",
            "newton_iterations_per_s": 1.0 / pdt,
            "sample": "%s: %d tet10 / %d dof, %d Newton iterations (SuperLU back-substitution + update_stress_load); one-time "
                      "calcGSM %.1fs and factorisation %.1fs excluded" % (pc["name"], ne, 3 * nn, iters, p_asm, p_fac),
            "solver_note": "scipy SuperLU (symmetric mode, 1 core) stands in for the reference's CHOLMOD (sksparse not installed); "
                           "CHOLMOD's back substitution is probably faster, so this CPU figure is likely lower than the reference's own",
            "same_mesh": {
                "workload": "%s: %d tet10 / %d dof" % (pc["name"], ne, 3 * nn),
                "cpu_newton_iterations_per_s": 1.0 / pdt, "cpu_cores": 1, "cpu_kind": "port",
                "cpu_one_time_s": {"calcGSM": p_asm, "factorisation": p_fac},
                "gpu_newton_iterations_per_s": gpu["newton_iterations_per_s"],
                "gpu_sample": "whole analysis (%d steps, %d Newton iterations, %d linear solves, %.1f PCG iterations per solve) incl. "
                              "assembly and multigrid set-up: %.2f s" % (gpu["steps"], gpu["newton_iterations"], gpu["linear_solves"],
                                                                       gpu["pcg_iterations_per_solve"], gpu["analysis_s"]),
                "gpu_over_cpu_newton_iterations_per_s": gpu["newton_iterations_per_s"] * pdt,
                "gpu_whole_analysis_s": gpu["analysis_s"],
                "cpu_whole_analysis_estimate_s": p_asm + p_fac + gpu["newton_iterations"] * pdt,
                "collapse_load_factor_gpu": gpu["collapse_load_factor"]}}


def precond_label(c):
    pi = c.precond_info()
    if pi["mode"] != "mg":
        return "block-Jacobi 3x3"
    form = {0: "multiplicative V-cycle (FP64 smoother SpMVs)", 1: "additive level-0 correction",
            2: "multiplicative V(1,1), smoother SpMVs on a 16-bit fixed-point copy of the matrix"}.get(pi.get("cycle_form", 1), "")
    return ("p-multigrid, %s: tet10 (block-Jacobi smoothing) -> tet4 on %d corner nodes%s" % (form, pi["coarse_nodes"], pi.get("level2", "")))


def collapse_run(case, warmup, steps):
    """The headline workload on one GPU: GpuModel of `case` + solver.collapse_run (see there).  Returns its dict extended with the
    model and the pattern / upload time."""
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel, collapse_run as run
    t0 = time.perf_counter()
    model = GpuModel(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                     Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    t_model = time.perf_counter() - t0
    out = run(model, case["ctl"], warmup, steps, rtol=RTOL)
    out.update(model=model, pattern_and_upload_s=t_model)
    return out


def cube_run(cells, warmup, steps, with_gnly=True):
    """BASELINE configs[2] geometry (synthetic cube, 1 M tet10) with the plasticity of the real analysis, iterating on one load
    step without a convergence test (error_max = 0): the headline of rounds 1-2, kept as a secondary object for continuity.  Its
    PCG count depends on how long the loop has run (start vectors from the solution history), which is why it is no longer `value`."""
    from fcvm_hip import meshgen
    from fcvm_hip.loads import Loads
    from fcvm_hip.solver import GpuModel, RiksDriver
    case = meshgen.cube_case(cells, l=100.0, traction=TRACTION)
    model = GpuModel(case["elNodes"], case["nocoord"], MATERIAL, case["fix"], Loads(ufaces=case["ufaces"], uloads=case["uloads"]))
    c = model.ctx
    st = {}
    drv = RiksDriver(model, CTL, rtol=RTOL, stats=st)
    drv.start_step()
    for _ in range(warmup):
        drv.iteration()
    c.sync()
    s0 = dict(st)
    t0 = time.perf_counter()
    for _ in range(steps):
        drv.iteration()
    c.sync()
    dt = time.perf_counter() - t0
    n_spmv = st["spmv_launches"] - s0["spmv_launches"]
    out = {"workload": "synthetic cube %d^3 cells = %d tet10, %d dof, one load step iterated %d + %d times (error_max = 0)" %
                       (cells, model.ne, model.ndof, warmup, steps),
           "ms_per_newton_iteration": dt / steps * 1e3, "pcg_iterations_per_newton_iteration": (st["pcg_iterations"] - s0["pcg_iterations"]) / steps,
           "spmv_avg_launch_ms": (st["spmv_ms"] - s0["spmv_ms"]) / max(n_spmv, 1), "pattern_host_s": model.t_pattern}
    del drv
    gnly = None if not with_gnly else gnly_run(model, cells)
    c.close()
    return out, gnly


def gnly_run(model, cells, warmup=1, steps=6):
    """The geometrically non-linear path (gnl = GNLY, fcVM.py:1351-1401) on the cube model: every Newton iteration re-assembles the
    tangent stiffness on the updated geometry (calcTSM: k_assemble_lds<TANGENT, SYM> straight into the SpMV's storage + the level-1
    operator), then solves K_t ue = f and K_t due = r with the new matrix -- as one lockstep pair (fcvm_solve2: every pass over the
    matrix serves both), or, when the previous tangent's ue still solves the new system to the tolerance, due alone from the carried
    start directions.  Timed with a stream synchronisation around every calc_gsm (two extra host waits per iteration)."""
    from fcvm_hip.solver import RiksDriver
    c = model.ctx
    st = {}
    drv = RiksDriver(model, dict(CTL, gnl="GNLY", maxImp="0.0", buckling=False), rtol=RTOL, stats=st)
    drv.start_step()
    orig, t_asm, n_asm = model.calc_gsm, [0.0], [0]

    def timed_gsm(*a, **kw):
        c.sync()
        t0 = time.perf_counter()
        o = orig(*a, **kw)
        c.sync()
        t_asm[0] += time.perf_counter() - t0
        n_asm[0] += 1
        return o
    model.calc_gsm = timed_gsm
    for _ in range(warmup):
        drv.iteration()
    c.sync()
    s0, t_asm[0], n_asm[0] = dict(st), 0.0, 0
    t0 = time.perf_counter()
    for _ in range(steps):
        drv.iteration()
    c.sync()
    dt = time.perf_counter() - t0
    model.calc_gsm = orig
    d = lambda k: st.get(k, 0) - s0.get(k, 0)  # noqa: E731
    out = {"workload": "synthetic cube %d^3 cells = %d tet10, gnl = GNLY: %d Newton iterations after %d, tangent re-assembled in %d of them"
                       % (cells, model.ne, steps, warmup, n_asm[0]),
           "ms_per_newton_iteration": dt / steps * 1e3, "tangent_calc_gsm_ms": t_asm[0] / max(n_asm[0], 1) * 1e3,
           "solve_ms_per_newton_iteration": d("solve_s") / steps * 1e3, "linear_solves_per_newton_iteration": d("solves") / steps,
           "pcg_iterations_per_newton_iteration": d("pcg_iterations") / steps,
           "lockstep_pair_iterations_per_newton_iteration": d("paired_iterations") / steps,
           "start_vector_spmv_per_newton_iteration": d("projection_spmv") / steps,
           "reference": "fcVM.py:1351-1401 (calcTSM + two back substitutions with the new factorisation per iteration)"}
    del drv
    return out


def run_single(args):
    from fcvm_hip import _lib, examples
    if _lib.device_count() < 1:
        raise SystemExit("bench.py: no HIP device / libfcvm_hip.so -- there is no CPU fallback")
    t0 = time.perf_counter()
    case = examples.plate_with_hole(args.plate_h)
    t_mesh = time.perf_counter() - t0
    run = collapse_run(case, args.warmup, args.steps)
    model, drv, c = run["model"], run["driver"], run["model"].ctx
    if "region" not in run:
        raise SystemExit("bench.py: the analysis ended (maximum restarts) before %d + %d Newton iterations" % (args.warmup, args.steps))
    reg = run["region"]
    s0, s1, dt = reg["stats0"], reg["stats1"], reg["s"]
    info = c.pattern_info()
    ndof, ne = info["ndof"], model.ne
    label = precond_label(c)      # (before the block-Jacobi detour below: the label reads the state of the cycle that ran)
    placement = c.placement_info()
    # one-time work of an analysis, timed on its own after the run: the bare assembly (calcGSM on the GPU), and assembly + hierarchy
    t_pat = model.t_pattern
    model.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    c.sync()
    t0 = time.perf_counter()
    model.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    c.sync()
    t_asm_setup = time.perf_counter() - t0
    mode = c.precond_info()["mode"]
    c.set_precond("bj")
    model.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    c.sync()
    t0 = time.perf_counter()
    model.calc_gsm((0.0, 0.0, 0.0), want_x=False)
    c.sync()
    t_asm = time.perf_counter() - t0
    c.set_precond(mode)
    # the same SpMV launched back to back on its own (no V-cycle in between, no p.Ap partials): context for the in-solver average
    xs, ys = c.array(np.random.default_rng(0).standard_normal(ndof)), c.empty(ndof)
    for _ in range(3):
        c.spmv(xs, ys)
    c.sync()
    t1 = time.perf_counter()
    for _ in range(20):
        c.spmv(xs, ys)
    c.sync()
    standalone_ms = (time.perf_counter() - t1) / 20 * 1e3
    del xs, ys
    stream_gbs = c.membench(3 << 30)
    spmv_launches = s1["spmv_launches"] - s0["spmv_launches"]
    spmv_ms = s1["spmv_ms"] - s0["spmv_ms"]
    pcg_its = s1["pcg_iterations"] - s0["pcg_iterations"]
    solves = s1["solves"] - s0["solves"]
    # algorithmic bytes of one SpMV launch: every STORED 3x3 block once (72 B values + 4 B column), x read once, y written once.
    # The symmetric form of the window format stores one block of every pair K_ij, K_ji inside a group (0.70 x the blocks of
    # the matrix at 1 M elements) and forms both products of it: its stored blocks are what the kernel has to stream.
    winfo = c.spmv_info()
    sinfo = c.spmv_sym_info()
    sym = sinfo["built"] and sinfo["in_use"]
    stored_blocks = sinfo["blocks"] if sym else info["nnz_blocks"]
    alg_bytes = stored_blocks * 76 + 2 * ndof * 8
    avg_ms = spmv_ms / max(spmv_launches, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    plastic = int(drv.pgp.get().sum())
    layout = ("window-%s%dx64-split%d" % ("sym-" if sym else "", winfo["waves_per_group"], winfo["split"])) if winfo["window_format"] else "sliced-ell-v%d" % info["spmv_variant"]
    traffic, traffic_src = spmv_traffic(case["name"], layout)
    an = run["analysis"]
    out = {
        "metric": "elements*GP/s over Newton iterations (Newton iterations/s x elements x 4 GP), 1M tet10",
        "value": ne * 4 * args.steps / dt, "unit": "elements*GP/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d tet10, %d dof -- the reference's Plate_with_hole_Example (geometry, supports, pressure load, von Mises "
                               "sig_yield 100, control file: 10 arc-length steps, error_max 5e-3) on a mapped mesh of %d tet10; a step = one "
                               "Newton/Riks iteration of that collapse analysis (iterations %d..%d of it, start_step / end_step between them "
                               "included; %d of %d Gauss points plastic at the end of the region)" %
                               (case["name"], ne, ndof, ne, args.warmup + 1, args.warmup + args.steps, plastic, 4 * ne),
                   "pcg_rtol": RTOL, "preconditioner": label},
        "newton_iterations_per_s": args.steps / dt,
        "pcg_iterations_per_step": pcg_its / args.steps,
        "pcg_iterations_per_solve": pcg_its / max(solves, 1),
        "fine_spmv_per_step": (s1.get("spmv_total", 0) - s0.get("spmv_total", 0)) / args.steps,
        "assembly": {"ms": t_asm * 1e3, "elements_gp_per_s": ne * 4 / t_asm,
                     "route": ("k_assemble_lds<SYM>: rows leave LDS straight into the SpMV's storage (symmetric window form, its 16-bit copy, "
                               "diagonal blocks)" if sinfo.get("assembled_directly") else "sliced-ELL array + copy into the SpMV's storage")},
        "setup_s": {"mesh": t_mesh, "pattern_and_upload": run["pattern_and_upload_s"], "pattern_host": t_pat,
                    "assembly_plus_preconditioner_hierarchy": t_asm_setup,
                    # once per mesh, inside the first assembly: the PCG's result vector and the symmetric form's value array timed against fresh
                    # allocations on the assembled matrix, the fastest kept (ms of the SpMV with the first / the kept allocation, ms spent)
                    "placement_tuning": placement},
        "roofline": {"bound": "hbm",
                     "kernel": (("k_spmv_win_sym<true> (window-format 3x3-block SpMV on the symmetric form: one block per pair K_ij, K_ji inside a "
                                 "group, transposed products through LDS in 64-bit fixed point; x staged in LDS, + p.Ap partials)") if sym else
                                "k_spmv_win<true> (window-format 3x3-block SpMV, x staged in LDS, + p.Ap partials)" if winfo["window_format"]
                                else "k_spmv<true> (sliced-ELL 3x3-block SpMV + p.Ap partials)"),
                     "storage": layout, "spmv_storage_bytes": sinfo["bytes"] if sym else winfo["bytes"] if winfo["window_format"] else info["matrix_bytes"],
                     "stored_blocks": stored_blocks, "matrix_blocks": info["nnz_blocks"],
                     # the same launch priced with ALL blocks of the matrix (what a full-storage kernel would have to stream):
                     "full_matrix_equivalent_GBs": (info["nnz_blocks"] * 76 + 2 * ndof * 8) / (avg_ms * 1e-3) / 1e9,
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms,
                     "launches_timed": spmv_launches, "back_to_back_launch_ms": standalone_ms, "padded_matrix_bytes": info["matrix_bytes"],
                     # what this box's HBM delivers for a plain 16 B/lane streaming read of a fresh 3 GB buffer, measured right
                     # after the timed region (boxes of the pool differ: 5.1 .. 6.8 TB/s): context for `frac`
                     "stream_read_GBs_this_box": stream_gbs,
                     "csr_equivalent_GBs": (info["nnz_blocks"] * 9 * 12 + 2 * ndof * 8) / (avg_ms * 1e-3) / 1e9},
        # the whole analysis the timed region is cut from (first pass = what the reference's .out documents), wall clock from
        # calcGSM to the last step, set-up (assembly, hierarchy, elastic solve) included
        "north_star_workload": dict(an, elements_gp_per_s=an["newton_iterations"] * ne * 4 / an["analysis_s"], workload="%s: %d tet10, %d dof, control file of the reference example (10 Riks steps to collapse)"
                                                 % (case["name"], ne, ndof),
                                    mesh_s=t_mesh, pattern_and_upload_s=run["pattern_and_upload_s"],
                                    reference_out_load_factor_395_elements=0.232),
    }
    del drv, model, run
    c.close()
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(plate_h=args.cpu_plate_h)
    if not args.no_cube:
        out["configs2_cube"], gnly = cube_run(args.cells, 5, 20, with_gnly=not args.no_gnly)
        if gnly is not None:
            out["gnly"] = gnly
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cells", type=int, default=55, help="cells per cube edge (55 -> 998,250 tet10)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-plate-h", type=float, default=6.0,
                    help="mesh size of the CPU sample / same-mesh comparison on Plate_with_hole (6.0 -> 16,830 tet10; 4.0 -> 48,600 = configs[1])")
    ap.add_argument("--no-cube", action="store_true", help="skip the secondary cube object and the GNLY object measured on its model (N = 1 only)")
    ap.add_argument("--no-gnly", action="store_true", help="skip the GNLY object (tangent re-assembly + pair of solves per Newton iteration)")
    ap.add_argument("--plate-h", type=float, default=1.45, help="mesh size of the headline workload (1.45 -> 1,006,434 tet10)")
    args = ap.parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run) BEFORE anything in this process touches the GPU, and pass their exit code on
        import socket
        import subprocess
        if os.environ.get("FCVM_DIST_SHARE_GPU") != "1":   # (rehearsal with all ranks on GPU 0 over gloo: this process stays off the GPU
            #                                                   altogether -- a GPU box admits six processes on its card, ranks included)
            import torch
            have = torch.cuda.device_count()   # counting devices does not initialise HIP
            if have < args.gpus:
                raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) visible" % (args.gpus, have))
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        if os.environ.get("FCVM_DIST_SHARE_GPU") == "1":
            # rehearsal: the ranks as plain children with the launcher's environment (torch.distributed.run itself opens the GPU, which
            # would be a seventh process on the card next to six ranks)
            procs = []
            for r in range(args.gpus):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
            rcs = [p.wait() for p in procs]
            raise SystemExit(max(rcs, key=abs))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(world_env) if world_env is not None else 1
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE=%d; launch with `python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node %d bench.py --gpus %d ...` or plain `python bench.py --gpus %d`"
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    if world > 1:
        from fcvm_hip import dist_bench
        dist_bench.run(args, CTL, MATERIAL, TRACTION, RTOL, HBM_PEAK_GBS)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
