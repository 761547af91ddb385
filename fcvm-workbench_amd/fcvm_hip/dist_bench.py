"""N > 1 leg of bench.py: one process per GPU (torch.distributed.run), weak scaling.  `run`: the headline workload of N = 1 -- the
reference's Plate_with_hole_Example with its control file -- on a mesh of N x 1 M tet10, elements cut into N ranges along a Morton
curve (dist.partition_model); the timed region is Newton iterations W+1 .. W+K of the collapse analysis, exactly as for N = 1, with a
barrier in its clock reads.  FCVM_DIST_WORKLOAD=column keeps the workload of rounds 1-2 (`build_rank_model`, `_run_column`: a column of
N stacked cubes, one slab of cells^3 cells per rank, base clamped, traction on the top face of the last slab), which the GPU tests of
tests/test_gpu_dist.py also use."""
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import meshgen
from .dist import DistBackend, DistModel, HipLocalOps, Partition, TVec
from .loads import Loads, external_loads
from .solver import RiksDriver


def build_rank_model(rank, world, cells, material, traction, device_index, backend_device, length=100.0, group=None):
    c = cells
    el, xyz = meshgen.cube_tet10(c, c, c, length, length, length)
    mx = my = 2 * c + 1
    el_g = el + rank * (2 * c) * mx * my  # z-slowest lattice numbering: slab r is a pure offset
    xyz = xyz.copy()
    xyz[:, 2] += rank * length
    part = Partition(el_g, group=group)
    assert part.nn == len(xyz) and np.array_equal(part.elNodes, el)
    fix = {}
    if rank == 0:
        for nd in meshgen.nodes_on_plane(xyz, 2, 0.0):
            for k in range(3):
                fix[3 * int(nd) + k] = 0.0
    if rank == world - 1:
        faces = meshgen.boundary_faces_on_plane(el, xyz, 2, world * length)
        # the shear part of the traction is scaled by 1/world: the column gets taller with the rank count and the
        # bending stress at the clamped base (which drives the plastic zone) would grow with it otherwise
        tr = np.asarray(traction, float) * np.array([1.0 / world, 1.0 / world, 1.0])
        loads = Loads(ufaces=faces, uloads=np.tile(tr, (len(faces), 1)))
    else:
        loads = Loads()
    ops = HipLocalOps(part, xyz, material, fix, device_index, element_offset=rank * len(el))
    be = DistBackend(part, ops, backend_device)
    if os.environ.get("FCVM_PRECOND", "mg") == "mg":
        be.enable_multigrid(xyz)
    model = DistModel(part, be, xyz, material, fix, loads, external_loads)
    return model, be, ops


def _shared_case(build, rank, world):
    """The global model of the weak-scaling run, built ONCE per node: rank 0 generates it (numpy, tens of seconds at 8 M tet10)
    and parks the arrays in a PRIVATE directory under /dev/shm (tempfile.mkdtemp: mode 0700, unpredictable name -- nobody else on
    a shared node can plant links there), the other ranks map them after the broadcast of its path (eight copies of the generator
    running side by side took several times as long and eight times the memory).  A failure of rank 0's build travels in the same
    broadcast, so the other ranks stop at once instead of sitting out the collective's time limit.  Single node (bench.py's
    contract); FCVM_DIST_SHARED_MESH=0: every rank builds its own."""
    if world == 1 or os.environ.get("FCVM_DIST_SHARED_MESH", "1") == "0" or not os.path.isdir("/dev/shm"):
        return build()
    import shutil
    import tempfile
    keys = ("elNodes", "nocoord", "material", "ufaces", "uloads")
    msg = [None]            # rank 0 -> all: ("dir", path) | ("own", None) | ("error", text)
    case, base = None, None
    if rank == 0:
        try:
            case = build()
        except BaseException as ex:       # noqa: B902 -- reported to every rank, then raised again below
            msg = [("error", "%s: %s" % (type(ex).__name__, ex))]
            err = ex
        else:
            try:
                base = tempfile.mkdtemp(prefix="fcvm_bench_", dir="/dev/shm")
                fd = np.fromiter(case["fix"].keys(), dtype=np.int64, count=len(case["fix"]))
                fv = np.fromiter(case["fix"].values(), dtype=np.float64, count=len(case["fix"]))
                for k in keys:
                    np.save(os.path.join(base, k + ".npy"), np.ascontiguousarray(case[k]))
                np.save(os.path.join(base, "fixdof.npy"), fd)
                np.save(os.path.join(base, "fixval.npy"), fv)
                with open(os.path.join(base, "meta.json"), "w") as fh:
                    json.dump({"ctl": case["ctl"], "name": case["name"]}, fh)
                msg = [("dir", base)]
            except OSError:       # /dev/shm not writable or full: every rank builds its own after all
                if base:
                    shutil.rmtree(base, ignore_errors=True)
                msg = [("own", None)]
    dist.broadcast_object_list(msg, src=0)     # (also the barrier behind which the files are complete)
    kind, val = msg[0]
    if kind == "error":
        if rank == 0:
            raise err
        raise SystemExit("bench.py: rank 0 could not build the global mesh (%s)" % val)
    if kind == "own":
        return case if rank == 0 else build()
    try:
        if rank != 0:
            case = {k: np.load(os.path.join(val, k + ".npy"), mmap_mode="r") for k in keys}
            fd, fv = np.load(os.path.join(val, "fixdof.npy")), np.load(os.path.join(val, "fixval.npy"))
            case["fix"] = {int(d): float(v) for d, v in zip(fd, fv)}
            with open(os.path.join(val, "meta.json")) as fh:
                case.update(json.load(fh))
            case["material"] = np.array(case["material"])
    finally:
        dist.barrier()     # everybody has mapped the files: the names can go (the mappings stay valid)
        if rank == 0:
            shutil.rmtree(val, ignore_errors=True)
    return case


def run(args, ctl, material, traction, rtol, hbm_peak):
    """N > 1: the headline workload of N = 1 -- the reference's Plate_with_hole_Example, its control file, Newton iterations
    warmup+1 .. warmup+steps of the collapse analysis -- on a mesh of N x (the one-GPU size) elements (weak scaling: h =
    plate_h / N^(1/3)), cut into N element ranges along a Morton curve (partition_model).  FCVM_DIST_WORKLOAD=column runs
    the stacked-cube column of rounds 1-2 instead (ctl / material / traction are that workload's)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("FCVM_DIST_DEBUG"):   # where does a stuck rank stand? tracebacks of all threads every N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FCVM_DIST_DEBUG"]), repeat=True)
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    share_gpu = os.environ.get("FCVM_DIST_SHARE_GPU") == "1"  # rehearsal: all ranks on GPU 0 over gloo
    if share_gpu:   # several processes on one GPU: one hardware queue each, or the scheduler time-slices them (see tests/conftest.py)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
    dev_index = 0 if share_gpu else local
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = os.environ.get("FCVM_DIST_BACKEND", "gloo" if share_gpu else "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    import datetime
    # a stuck collective should end the run with an error well inside the caller's time limit, not sit out the default 10 min
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=4))
    try:
        if os.environ.get("FCVM_DIST_WORKLOAD", "plate") == "column":
            return _run_column(args, ctl, material, traction, rtol, hbm_peak, rank, world, dev_index, device, backend)
        from . import examples
        from .dist import partition_model
        from .solver import collapse_run
        t0 = time.perf_counter()
        case = _shared_case(lambda: examples.plate_with_hole(args.plate_h / world ** (1.0 / 3.0)), rank, world)
        t_mesh = time.perf_counter() - t0
        t0 = time.perf_counter()
        model = partition_model(case["elNodes"], case["nocoord"], case["material"], case["fix"],
                                Loads(ufaces=case["ufaces"], uloads=case["uloads"]), device_index=dev_index, device=device,
                                precond=os.environ.get("FCVM_PRECOND", "mg"))
        be, ops = model.ctx, model.ctx.ops
        t_part = time.perf_counter() - t0

        def sync():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        run_ = collapse_run(model, case["ctl"], args.warmup, args.steps, rtol=rtol, sync=sync)
        if "region" not in run_:
            raise SystemExit("bench.py: the analysis ended before %d + %d Newton iterations" % (args.warmup, args.steps))
        reg = run_["region"]
        tmax = torch.tensor([reg["s"]], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        s0, s1 = reg["stats0"], reg["stats1"]
        # roofline object: the IN-SOLVER launches of rank 0's fine SpMV kernel inside the timed region (HIP events on the stream the kernel
        # runs on, around the kernel alone -- the shared-node exchange that follows is outside; DistBackend._solve_single_reduction), and
        # the same kernel back to back afterwards for context
        x, y = be.array(np.random.default_rng(rank).standard_normal(3 * model.nn)), be.empty(3 * model.nn)
        for _ in range(3):
            ops.spmv(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            ops.spmv(x, y)
        e1.record()
        torch.cuda.synchronize()
        standalone_ms = e0.elapsed_time(e1) / reps
        launches = s1.get("spmv_launches", 0) - s0.get("spmv_launches", 0)
        avg_ms = (s1.get("spmv_ms", 0.0) - s0.get("spmv_ms", 0.0)) / launches if launches > 0 else standalone_ms
        info = ops.c.pattern_info()
        sinfo = ops.c.spmv_sym_info()      # the symmetric form stores (and streams) fewer blocks than the matrix has
        alg = (sinfo["blocks"] if sinfo["built"] and sinfo["in_use"] else info["nnz_blocks"]) * 76 + 2 * info["ndof"] * 8
        ne_total = len(case["elNodes"])
        pcg = s1["pcg_iterations"] - s0["pcg_iterations"]
        solves = s1["solves"] - s0["solves"]
        shared = int(sum(len(ix) for ix in model.part.shared.values()))
        sh_all = [None] * world
        dist.all_gather_object(sh_all, (model.ne, model.nn, shared))
        if rank == 0:
            achieved = alg / (avg_ms * 1e-3) / 1e9
            an = run_["analysis"]
            out = {
                "metric": "elements*GP/s over Newton iterations (Newton iterations/s x elements x 4 GP), 1M tet10 per GPU",
                "value": ne_total * 4 * args.steps / dt, "unit": "elements*GP/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "%s: %d tet10 (%d per GPU on average), %d dof -- the reference's Plate_with_hole_Example with its control "
                                       "file on a mesh of %d x the one-GPU size, elements cut into %d ranges along a Morton curve; a step = one "
                                       "Newton/Riks iteration of the collapse analysis (iterations %d..%d)" %
                                       (case["name"], ne_total, ne_total // world, 3 * len(case["nocoord"]), world, world, args.warmup + 1,
                                        args.warmup + args.steps),
                           "pcg_rtol": rtol,
                           "preconditioner": (("distributed p-multigrid, %s (tet10 -> tet4 -> %d aggregates x rigid-body modes, replicated on "
                                               "every rank: %s), single-reduction CG" %
                                               ({2: "multiplicative V(1,1) with 16-bit smoother products per rank", 1: "additive cycle",
                                                 0: "multiplicative Chebyshev cycle"}[be.mg.form], be.mg.nagg,
                                                ("sparse 6 x 6 blocks with block-Jacobi Chebyshev sweeps -> %d super-aggregates, dense inverse"
                                                 % be.mg.l2_info["super_aggregates"]) if be.mg.l2_info["sparse"] else "dense inverse"))
                                              if be.mg is not None else
                                              ("the one-GPU p-multigrid PCG of the library (single rank: nothing to exchange)" if be.local_solver
                                               else "block-Jacobi 3x3")), "parallelism": "element-partition x%d" % world,
                           "partition": {"elements_nodes_interface_nodes_per_rank": sh_all},
                           "exchange": "point-to-point shared-node sums after every SpMV and the restriction, scalar and aggregate-level "
                                       "all-reduces (%s); per PCG iteration: %s" % (backend, run_["stats"].get("comm_per_iteration", "n/a"))},
                "newton_iterations_per_s": args.steps / dt, "pcg_iterations_per_step": pcg / args.steps,
                "pcg_iterations_per_solve": pcg / max(solves, 1),
                "setup_s": {"mesh": t_mesh, "partition_pattern_upload": t_part, "inside_analysis": run_["setup_s"]},
                "north_star_workload": dict(an, elements_gp_per_s=an["newton_iterations"] * ne_total * 4 / an["analysis_s"]),
                "roofline": {"bound": "hbm", "kernel": "k_spmv_win_sym<true> on rank 0's matrix (%d elements), launches inside the PCG of the timed "
                                                       "region" % model.ne,
                             "achieved": achieved, "peak": hbm_peak, "unit": "GB/s", "frac": achieved / hbm_peak, "traffic": None,
                             "traffic_note": "no PMC pass over a rank's partition; the same kernel and layout on the N = 1 workload move 1.039 x "
                                             "their algorithmic bytes (profiles/r05_plate_spmv_sym_traffic_pmc.txt)",
                             "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms, "launches_timed": launches,
                             "back_to_back_launch_ms": standalone_ms},
                # HIP memory in use on rank 0's device at the end (all ranks' when they share the GPU in a rehearsal) / the peak torch saw
                "device_memory_GB": {"in_use_on_device": (torch.cuda.mem_get_info(dev_index)[1] - torch.cuda.mem_get_info(dev_index)[0]) / 1e9,
                                     "ranks_on_this_device": world if share_gpu else 1,
                                     "torch_peak_this_rank": torch.cuda.max_memory_allocated(dev_index) / 1e9},
                "aggregate_level": be.mg.l2_info if be.mg is not None else None,
                "multigrid_setup_s": be.mg.timing if be.mg is not None else None,
            }
        # cpu_baseline: rank 0's host, the same bounded sample as for N = 1 (the other ranks wait in the barrier below)
        if rank == 0 and not args.no_cpu:
            import bench
            out["cpu_baseline"] = bench.cpu_baseline(plate_h=args.cpu_plate_h)
        dist.barrier()
        if rank == 0:
            print(json.dumps(out))
    finally:
        dist.destroy_process_group()


def _run_column(args, ctl, material, traction, rtol, hbm_peak, rank, world, dev_index, device, backend):
    if True:
        model, be, ops = build_rank_model(rank, world, args.cells, material, traction, dev_index, device)
        stats = {}
        drv = RiksDriver(model, ctl, rtol=rtol, stats=stats)
        drv.start_step()
        for _ in range(args.warmup):
            drv.iteration()
        dist.barrier()
        torch.cuda.synchronize()
        s0 = dict(stats)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            drv.iteration()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # local SpMV timing for the roofline object (HIP events on the stream the kernel runs on)
        x, y = be.array(np.random.default_rng(rank).standard_normal(3 * model.nn)), be.empty(3 * model.nn)
        for _ in range(3):
            ops.spmv(x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            ops.spmv(x, y)
        e1.record()
        torch.cuda.synchronize()
        avg_ms = e0.elapsed_time(e1) / reps
        info = ops.c.pattern_info()
        sinfo = ops.c.spmv_sym_info()      # the symmetric form stores (and streams) fewer blocks than the matrix has
        alg = (sinfo["blocks"] if sinfo["built"] and sinfo["in_use"] else info["nnz_blocks"]) * 76 + 2 * info["ndof"] * 8
        ne_total = model.ne * world
        pcg = stats["pcg_iterations"] - s0["pcg_iterations"]
        if rank == 0:
            achieved = alg / (avg_ms * 1e-3) / 1e9
            out = {
                "metric": "elements*GP/s over Newton iterations (Newton iterations/s x elements x 4 GP), 1M tet10 per GPU",
                "value": ne_total * 4 * args.steps / dt, "unit": "elements*GP/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "column of %d cubes of %d^3 cells = %d tet10 (%d per GPU), element range partition "
                                       "in z-slabs, clamped base, top traction, von Mises plasticity, one Newton/Riks "
                                       "iteration per step" % (world, args.cells, ne_total, model.ne),
                           "pcg_rtol": rtol,
                           "preconditioner": (("distributed p-multigrid V-cycle (tet10 -> tet4 -> %d aggregates x rigid-body modes, dense inverse "
                                               "split by rows over the ranks)" % be.mg.nagg) if be.mg is not None else
                                              ("the one-GPU p-multigrid PCG of the library (single rank: nothing to exchange)" if be.local_solver
                                               else "block-Jacobi 3x3")), "parallelism": "element-partition x%d" % world,
                           "exchange": "point-to-point shared-node sums per SpMV (1 fine + 5 coarse per PCG iteration), 2 scalar and 2 aggregate-level all-reduces (%s)" % backend},
                "newton_iterations_per_s": args.steps / dt, "pcg_iterations_per_step": pcg / args.steps,
                "roofline": {"bound": "hbm", "kernel": "k_spmv (rank 0, local matrix)", "achieved": achieved, "peak": hbm_peak,
                             "unit": "GB/s", "frac": achieved / hbm_peak, "traffic": None,
                             "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms},
            }
            print(json.dumps(out))
