#!/usr/bin/env python3
"""CPU (oracle port: SuperLU back substitution + update_stress_load, one core) against GPU (whole collapse analysis through the
C-ABI) Newton iterations per second on ONE mesh per row, the reference's Plate_with_hole_Example at several sizes
(VERDICT round 2, item 7).  Writes a JSON table; bench.py's cpu_baseline.same_mesh is the first row of it.

  python tools/cpu_gpu_table.py --h 6.0,4.0 --out profiles/r03_cpu_gpu_table.json
"""
import argparse
import json
import os
import resource
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", default="6.0,4.0")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--merge", default=None, help="an existing table to extend")
    a = ap.parse_args()
    from fcvm_hip import examples
    from oracle import oracle as orc
    rows = json.load(open(a.merge))["rows"] if a.merge and os.path.exists(a.merge) else []
    for h in [float(v) for v in a.h.split(",")]:
        pc = examples.plate_with_hole(h)
        ne, nn = len(pc["elNodes"]), len(pc["nocoord"])
        t0 = time.perf_counter()
        pdt, p_asm, p_fac = bench._cpu_newton_iterations(orc, pc["elNodes"], pc["nocoord"], pc["material"], pc["fix"],
                                                         orc.Loads(ufaces=pc["ufaces"], uloads=pc["uloads"]), a.iters,
                                                         float(pc["ctl"]["sig_yield"]))
        cpu_wall = time.perf_counter() - t0
        gpu = bench.collapse_run(pc, 0, None)["analysis"]
        row = {"mesh": pc["name"], "tet10": ne, "dof": 3 * nn,
               "cpu": {"kind": "port", "solver": "scipy SuperLU, symmetric mode, 1 core (stand-in for the reference's CHOLMOD)",
                       "calcGSM_s": p_asm, "factorisation_s": p_fac, "s_per_newton_iteration": pdt,
                       "newton_iterations_per_s": 1.0 / pdt, "peak_rss_GB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6,
                       "wall_s": cpu_wall},
               "gpu": {"whole_analysis_s": gpu["analysis_s"], "newton_iterations": gpu["newton_iterations"],
                       "newton_iterations_per_s": gpu["newton_iterations_per_s"], "pcg_iterations_per_solve": gpu["pcg_iterations_per_solve"],
                       "collapse_load_factor": gpu["collapse_load_factor"]},
               "gpu_over_cpu_newton_iterations_per_s": gpu["newton_iterations_per_s"] * pdt,
               "cpu_whole_analysis_estimate_s": p_asm + p_fac + gpu["newton_iterations"] * pdt}
        print(json.dumps(row), flush=True)
        rows = [r for r in rows if r["mesh"] != row["mesh"]] + [row]
        if a.out:
            rows.sort(key=lambda r: r["tet10"])
            json.dump({"what": "Plate_with_hole_Example, shipped control file, one mesh per row for both sides; CPU Newton iteration = back "
                               "substitution + update_stress_load with the factorisation outside (the reference factorises once, "
                               "fcVM.py:1121); GPU = the whole collapse analysis incl. assembly and multigrid set-up", "rows": rows},
                      open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
