"""Loader for the CPU restatement (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Parity unpinned (see md_oracle.cpp header): the restatement is pinned by
analytic known answers, not by reference output.

The oracle exports the product's C ABI under the `orc_` prefix, so the product's own host
wrapper (chemlab_amd.engine.Engine) drives it unchanged.
"""
import ctypes as C
import os
import subprocess

from chemlab_amd import _capi
from chemlab_amd.engine import Engine

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

_EXTRA = {"create": (C.c_void_p, []), "compute_forces": (C.c_int, [C.c_void_p])}
_api = None


def build(force=False):
    src = os.path.join(HERE, "md_oracle.cpp")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB


def api():
    global _api
    if _api is None:
        build()
        _api = _capi.bind(C.CDLL(LIB), "orc_", _EXTRA)
    return _api


class OracleEngine(Engine):
    def __init__(self):
        a = api()
        super().__init__(api=a, ctx=a.create(), precision=64)

    def compute_forces(self):
        self._ck(self.api.compute_forces(self.ctx))
