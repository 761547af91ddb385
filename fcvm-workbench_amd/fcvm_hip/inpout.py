"""The reference's text files: control file reader (.inp) and result table writer (.out).

Both formats are defined by the macro that drives the reference solver
(reference: source code/fcVM.FCMacro:70-96 reads the 21-line .inp; :190-243 writes the .out).
"""
INP_KEYS = ("sig_yield", "grav_x", "grav_y", "grav_z", "nstep", "iterat_max", "error_max", "relax", "scale_re",
            "scale_up", "scale_dn", "disp_output", "ultimate_strain", "Et_E", "target_LF", "csr_option",
            "averaged_option", "gnl", "maxImp", "ev1", "ev2")
_INT = {"nstep", "iterat_max"}
_STR = {"disp_output", "csr_option", "averaged_option", "gnl", "maxImp", "ev1", "ev2"}


_MIN_LINES = 15  # up to target_LF every field is numeric or used unconditionally


def parse_inp_lines(lines):
    """Older control files stop after `averaged_option` (17 lines, e.g. VM_Uniaxial_Tension_Example.inp):
    the macro's readline() then returns "" for gnl / maxImp / ev1 / ev2 (fcVM.FCMacro:91-94), which
    selects the geometrically linear analysis.  A file that ends inside the numeric block makes the
    macro's float("") raise, so that is an error here too."""
    t = [s.strip() for s in lines if s.strip() != ""]
    if len(t) < _MIN_LINES:
        raise ValueError("control file needs at least %d lines, got %d" % (_MIN_LINES, len(t)))
    t = t + [""] * (len(INP_KEYS) - len(t))
    ctl = {}
    for k, v in zip(INP_KEYS, t):
        ctl[k] = v if k in _STR else (int(v) if k in _INT else float(v))
    return ctl


def read_inp(path):
    with open(path, encoding="utf8") as f:
        return parse_inp_lines(f.readlines())


def write_inp(path, ctl):
    with open(path, "w", encoding="utf8") as f:
        for k in INP_KEYS:
            f.write("%s\n" % ctl[k])


_SEP = "\n=========================================================================================================================\n\n"


def write_out(path, name, ne, nn, gnl, nstep, eigenval, loadsum, crip, x, lout, un, peeqplot, pplot, svmplot,
              triaxplot, ecrplot, csrplot):
    """fcVM.FCMacro:190-243.  `nn` is the node count (the reference prints it under
    "No. of Degrees of freedom")."""
    with open(path, "w") as f:
        f.write("model name:{0: >50}\n".format(name))
        f.write("No. of elements:{0: >45}\n".format(ne))
        f.write("No. of Degrees of freedom:{0: >35}\n".format(nn))
        if gnl == "GNLY":
            if float(nstep) == 1.0:
                f.write("analysis type:{0: >47}\n".format("elastic buckling analysis"))
            else:
                f.write("analysis type:{0: >47}\n".format("elastic-plastic, geometric non-linear"))
            f.write("elastic buckling factors:{0: >36}\n".format(str(eigenval)))
        else:
            if float(nstep) == 1.0:
                f.write("analysis type: elastic\n")
            else:
                f.write("analysis type: elastic-plastic, geometric linear\n")
        f.write(_SEP)
        f.write("Sum of loads x-direction: {0: >15.2e}\n".format(loadsum[0]))
        f.write("Sum of loads y-direction: {0: >15.2e}\n".format(loadsum[1]))
        f.write("Sum of loads z-direction: {0: >15.2e}\n".format(loadsum[2]))
        f.write(_SEP)
        f.write("{0: >8}{1: >10}{2: >10}{3: >10}{4: >10}{5: >10}{6: >10}{7: >10}{8: >10}{9: >10}{10: >10}{11: >10}\n".format(
            "Gauss point", "x", "y", "z", "load", "disp", "peeq", "pressure", "svmises", "triax", "eps_cr", "csr_max"))
        # displacement-controlled runs that stop at the target load carry one load value less than result rows (the
        # reaction-force history is not extended in that branch, fcVM.py:1486-1510; the macro's loop over len(crip)
        # then raises IndexError): the complete rows are written
        for i in range(min(len(crip), len(lout))):
            f.write("{0: 11d}{1: >10.2e}{2: >10.2e}{3: >10.2e}{4: >10.2e}{5: >10.2e}{6: >10.2e}{7: >10.2e}{8: >10.2e}{9: >10.2e}{10: >10.2e}{11: >10.2e}\n".format(
                int(crip[i]), x[crip[i]][0], x[crip[i]][1], x[crip[i]][2], lout[i], un[i], peeqplot[i], pplot[i],
                svmplot[i], triaxplot[i], ecrplot[i], csrplot[i]))
        f.write(_SEP)
