"""GROMACS 7-column .xvg table -> ESPResSo++ 3-column `r e f` .pot table.
Behaviour of /root/reference/tools/convert_gromacs2espp.py:28-110: non-bonded tables have columns
(r, f, -f', g, -g', h, -h'), U = c6-weighted g + c12-weighted h (defaults 1,1); bonded tables have
(x, e, f).  The kind is taken from the FILE NAME like the reference does (:44-58): `_b<N>` bond,
`_a<N>` angle, `_d<N>` dihedral; angle and dihedral tables are in degrees and are converted to radians
(x -> radians(x), f -> f*180/pi, :73-75), keeping 0 < theta <= pi resp. -pi <= phi <= pi (:81-83);
the r = 0 row of distance tables is dropped; numbers are printed with %15.8g.  Reduced units (:69-72,97-99):
distances are divided by sigma, energies by epsilon, forces multiplied by sigma/epsilon (angles stay radians)."""
import math
import os
import re


def convert_table(gro_in, espp_out, sigma=1.0, epsilon=1.0, c6=1.0, c12=1.0):
    rows = []
    bonded = None
    with open(gro_in) as f:
        for line in f:
            t = line.strip()
            if not t or t[0] in "#@":
                continue
            cols = t.split()
            if bonded is None:
                bonded = len(cols) == 3
            rows.append([float(c) for c in cols])
    out = []
    base = os.path.basename(gro_in)
    angle = bool(bonded and re.match(r".*_a[0-9]+.*", base))
    dihedral = bool(bonded and not angle and re.match(r".*_d[0-9]+.*", base))
    for c in rows:
        r = c[0]
        if angle or dihedral:
            r = math.radians(r)
            e, f_ = c[1] / epsilon, c[2] * 180.0 / math.pi * sigma / epsilon
            if (angle and not (0 < r <= math.pi)) or (dihedral and not (-math.pi <= r <= math.pi)):
                continue
            out.append("%15.8g %15.8g %15.8g\n" % (r, e, f_))
            continue
        if r == 0.0:     # the singular first row is not usable by the interpolation
            continue
        if bonded:
            e, f_ = c[1], c[2]
        else:
            e = c6 * c[3] + c12 * c[5]
            f_ = c6 * c[4] + c12 * c[6]
        r, e, f_ = r / sigma, e / epsilon, f_ * sigma / epsilon
        out.append("%15.8g %15.8g %15.8g\n" % (r, e, f_))
    # written under a private name and renamed: with several ranks converting the same table at once (multi-process runs:
    # every rank sets the interactions up) a reader never sees half a file, and all writers produce the same bytes
    tmp = "%s.tmp%d" % (espp_out, os.getpid())
    with open(tmp, "w") as f:
        f.writelines(out)
    os.replace(tmp, espp_out)
    return len(out)
