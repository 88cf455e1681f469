"""chemlab_amd -- MI355X-native reactive-MD inner loop behind ChemLab's espressopp-facing API.

Layout (hot path only):
  csrc/        HIP kernels for gfx950 + the C ABI (libchem_mi355.so)
  _capi.py     ctypes binding of include/chem_mi355.h
  engine.py    one context == one GPU, numpy in / numpy out
  espp/        py3 `espressopp`-shaped shim used by the ChemLab driver logic
  workloads.py synthetic BASELINE configurations (C2..C5)
"""
__version__ = "0.1.0"
