"""Philox4x32-10 known-answer vectors (Random123 kat_vectors, Salmon et al. SC'11) pin the
counter-based streams shared by the oracle and the HIP path (include/chem_philox.h)."""
import ctypes as C

import numpy as np

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox_py(ctr, key):
    c = list(ctr)
    k = list(key)
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xffffffff, p1 & 0xffffffff,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xffffffff, p0 & 0xffffffff]
        k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
    return tuple(c)


def test_python_restatement_matches_kat():
    for ctr, key, out in KAT:
        assert philox_py(ctr, key) == out


def test_header_implementation_matches_kat(oracle_mod):
    lib = C.CDLL(oracle_mod.LIB)
    for ctr, key, out in KAT:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        lib.orc_philox4x32_10(c, k, o)
        assert tuple(o) == out


def test_random_blocks_match_python(oracle_mod):
    lib = C.CDLL(oracle_mod.LIB)
    rng = np.random.default_rng(0)
    for _ in range(50):
        ctr = tuple(int(x) for x in rng.integers(0, 2**32, 4))
        key = tuple(int(x) for x in rng.integers(0, 2**32, 2))
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        lib.orc_philox4x32_10(c, k, o)
        assert tuple(o) == philox_py(ctr, key)
