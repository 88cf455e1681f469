"""GROMACS 7-column .xvg table -> ESPResSo++ 3-column `r e f` .pot table.
Behaviour of /root/reference/tools/convert_gromacs2espp.py:28-110 for plain non-bonded/bonded
tables: columns (r, f, -f', g, -g', h, -h'); U = c6-weighted g + c12-weighted h (defaults 1,1),
the r = 0 row is dropped, numbers printed with %15.8g."""


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
    for c in rows:
        r = c[0]
        if r == 0.0:     # the singular first row is not usable by the interpolation
            continue
        if bonded:
            e, f_ = c[1], c[2]
        else:
            e = c6 * c[3] + c12 * c[5]
            f_ = c6 * c[4] + c12 * c[6]
        out.append("%15.8g %15.8g %15.8g\n" % (r, e, f_))
    with open(espp_out, "w") as f:
        f.writelines(out)
    return len(out)
