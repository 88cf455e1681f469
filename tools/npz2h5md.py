#!/usr/bin/env python3
"""<prefix>_traj.npz (written by the driver where h5py is missing) -> <prefix>_traj.h5 with the same dataset tree
(/particles/atoms/*/value, /connectivity/chem_bonds_<i>/value, ...).  Needs h5py:  python tools/npz2h5md.py run_traj.npz"""
import sys

import numpy as np


def convert(src, dst=None):
    import h5py
    dst = dst or src[:-4] + ".h5"
    with np.load(src) as z, h5py.File(dst, "w") as h5:
        for path in z.files:
            h5.create_dataset(path, data=z[path])
    return dst


if __name__ == "__main__":
    print(convert(*sys.argv[1:3]))
