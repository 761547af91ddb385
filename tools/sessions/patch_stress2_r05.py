"""The vertex-patch test's body N times in ONE process (the suite's one failure in eight runs did not come back in fresh processes)."""
import os, sys, traceback
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fcvm-workbench_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
from _pytest.monkeypatch import MonkeyPatch
import test_gpu_parity as T
fn = T.test_vertex_patch_smoother_is_the_additive_schwarz_operator_and_cuts_the_iterations
bad = 0
for k in range(int(sys.argv[1])):
    for name in ("plate", "embankment"):
        mp = MonkeyPatch()
        try:
            fn(mp, name)
            print("run %2d %-10s ok" % (k, name), flush=True)
        except Exception as e:
            bad += 1
            print("run %2d %-10s FAILED: %s" % (k, name, str(e)[:200]), flush=True)
            traceback.print_exc(limit=2)
        finally:
            mp.undo()
print("failures:", bad)
