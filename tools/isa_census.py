#!/usr/bin/env python3
"""Memory-op / instruction census of selected kernels in the gfx950 ISA (hipcc -S)."""
import re, subprocess, sys
src = "chemlab_amd/csrc/chem_api.hip"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "/tmp/chem.s", src])
s = open("/tmp/chem.s").read()
pat = sys.argv[1:] or ["k_pair_tilesIfLi4ELb0", "k_nlist_tilesIf", "k_pair_forceIfLi4ELb0"]
funcs = re.split(r"\n(?=_ZN4chem[^\n:]*: )", s)
for f in funcs:
    name = f.split(":")[0]
    if not any(p in name for p in pat):
        continue
    body = f.split(".end_amdhsa_kernel")[0] if ".end_amdhsa_kernel" in f else f
    body = body.split("s_endpgm")[0]
    ops = {}
    for line in body.split("\n")[1:]:
        t = line.strip().split(" ")[0]
        if t and not t.startswith((".", ";", "_")) and not t.endswith(":"):
            ops[t] = ops.get(t, 0) + 1
    mem = {k: v for k, v in ops.items() if k.startswith(("flat", "ds_", "global_", "scratch", "buffer"))}
    print(name[:60], "instr", sum(ops.values()), mem)
