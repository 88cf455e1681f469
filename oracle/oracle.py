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


LIB_OMP = os.path.join(HERE, "liboracle_omp.so")
_api_omp = None


def lscpu_cores():
    """(logical CPUs, physical cores) of the host as `lscpu` reports them (the core count goes into the log next to
    the CPU baseline); None where lscpu is missing."""
    import subprocess
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        return None
    kv = {}
    for line in txt.splitlines():
        if ":" in line:
            k, v = line.split(":", 1)
            kv[k.strip()] = v.strip()
    try:
        logical = int(kv["CPU(s)"])
        phys = int(kv["Core(s) per socket"]) * int(kv["Socket(s)"])
        return logical, phys
    except Exception:
        return None


def host_cores():
    """Cores this process may actually use: the affinity mask (what `nproc` prints) capped by the cgroup CPU
    quota when one is set (a GPU box shows every core of the host but grants a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / float(per) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    cap = os.environ.get("CHEM_CPU_BASELINE_CORES")   # explicit override
    if cap:
        n = max(1, int(cap))
    return n


def api_omp():
    """All-core build of the same restatement (-O3 -march=native -fopenmp).  -march=native does not
    travel between machines, so it is always (re)built on the host that is going to time it."""
    global _api_omp
    if _api_omp is None:
        stamp = LIB_OMP + ".host"
        here = "%s|%s" % (os.uname().nodename, _cpu_model())
        if not (os.path.exists(LIB_OMP) and os.path.exists(stamp) and open(stamp).read() == here
                and os.path.getmtime(LIB_OMP) >= os.path.getmtime(os.path.join(HERE, "md_oracle.cpp"))):
            subprocess.check_call(["make", "-C", HERE, "-B", "liboracle_omp.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            with open(stamp, "w") as f:
                f.write(here)
        _api_omp = _capi.bind(C.CDLL(LIB_OMP), "orc_", _EXTRA)
    return _api_omp


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "?"


class OracleEngine(Engine):
    def __init__(self, threads=0):
        """threads = 0: the scalar checker every parity test uses.  threads >= 1: the OpenMP build with
        that many threads (bench.py's cpu_baseline: 1 core and all cores)."""
        a = api() if threads <= 0 else api_omp()
        super().__init__(api=a, ctx=a.create(), precision=64)
        if threads >= 1:
            self.set_option("threads", threads)

    def compute_forces(self):
        self._ck(self.api.compute_forces(self.ctx))
