"""Which rank of a multi-process run writes files.

The reference is started as `mpirun -n N python start_simulation.py @params`; ESPResSo++'s controller model (pmi) runs the
script -- and with it every output statement -- once, the workers only execute the storage / integrator commands
(src/start_simulation.py:152-163 picks the node grid from MPI.COMM_WORLD.size).  This build is SPMD: `python -m
torch.distributed.run --nproc-per-node N -m chemlab_amd.start_simulation @params` runs the whole driver on every rank (the
host topology and every read-back are replicated or collective, so all ranks must make the same calls); only rank 0 writes.
Every writer of the package opens its files through `wopen`, which hands the other ranks the null device.
"""
import builtins
import os

_root = [True]


def set_root(flag):
    _root[0] = bool(flag)


def is_root():
    return _root[0]


def wopen(path, mode="r", *a, **k):
    """open() for output files: the real file on the writing rank, os.devnull elsewhere (reads pass through)."""
    if not _root[0] and any(c in mode for c in "wax+"):
        return builtins.open(os.devnull, "w" if "b" not in mode else "wb")
    return builtins.open(path, mode, *a, **k)
