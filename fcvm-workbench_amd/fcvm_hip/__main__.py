"""Headless run of a saved FreeCAD document, the way the reference's macro runs the active document:

    PYTHONPATH=fcvm-workbench_amd python -m fcvm_hip <model.FCStd> [--control DIR] [--output DIR] [--device N]

reads DIR/<label>.inp (the 21-line control file of fcVM.FCMacro:70-96; <label> = the document's label, as in the
reference) and writes <label>.out / <label>.vtk in the reference's formats.  Needs libfcvm_hip.so and an MI355X; there
is no CPU fallback."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fcVM  # noqa: E402
from fcvm_hip import fcstd  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m fcvm_hip", description=__doc__.splitlines()[0])
    ap.add_argument("document", help=".FCStd file with an embedded 2nd-order tetrahedral mesh")
    ap.add_argument("--control", default=None, help="directory of <label>.inp (default: next to the document)")
    ap.add_argument("--output", default=None, help="directory for <label>.out / .vtk (default: the control directory)")
    ap.add_argument("--name", default=None, help="model name (default: the document's label)")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    control = a.control or os.path.dirname(os.path.abspath(a.document))
    output = a.output or control
    name = a.name or fcstd.load_model(a.document)["name"]
    fcVM.settings["device"] = a.device
    res = fcVM.run_fcstd(a.document, control, output, name=name)
    print("%s: %d load steps, final load factor %.4g, max displacement %.4g -> %s" % (
        name, len(res["lout"]) - 1, res["lout"][-1], res["un"][-1], os.path.join(output, name + ".out")))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
