import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd"))
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.lib()
    return oracle


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.load(open(os.path.join(GOLDEN, name + ".json")))
    return g, meta


@pytest.fixture(scope="session")
def embankment():
    return load_golden("embankment")


@pytest.fixture(scope="session")
def gpu_ctx():
    from fcvm_hip import _lib
    from fcvm_hip.device import Context
    if _lib.device_count() < 1:
        pytest.fail("gpu tests need a HIP device and libfcvm_hip.so (no CPU fallback)")
    return Context(0)


def small_case(n=3, prescribed=True):
    """cube with clamped bottom; a few prescribed non-zero displacements on the top corner nodes."""
    from fcvm_hip import meshgen
    case = meshgen.cube_case(n, l=10.0, traction=(1.0, 0.5, 8.0))
    if prescribed:
        top = meshgen.nodes_on_plane(case["nocoord"], 2, 10.0)
        for nd in top[:3]:
            case["fix"][3 * int(nd) + 2] = 0.01
            case["fix"][3 * int(nd)] = -0.002
    case["material"] = np.array([210000.0, 0.3, 7.8e-6])
    return case


def spawn_sharing_gpu(fn, args, nprocs):
    """mp.spawn of `nprocs` ranks that all use GPU 0 (gloo control plane).  Every HIP process opens up to four hardware
    queues by default; from three processes on the hardware scheduler time-slices them and every host<->device
    synchronisation of the host-staged exchange costs milliseconds (measured: a 136-iteration solve on three ranks 9.1 s,
    1.5 s with one queue per process; profiles/r02q_three_ranks_hw_queues.txt).  One queue per rank for the rehearsal."""
    import os
    import torch.multiprocessing as mp
    old = os.environ.get("GPU_MAX_HW_QUEUES")
    os.environ["GPU_MAX_HW_QUEUES"] = "1"
    try:
        mp.spawn(fn, args=args, nprocs=nprocs, join=True)
    finally:
        if old is None:
            del os.environ["GPU_MAX_HW_QUEUES"]
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = old
