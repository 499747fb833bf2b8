"""Pins the oracle's Morton encoder and interval Horner against the reference's own
headers: src/detail/mortonND_LUT.h (MortonNDLutEncoder<4,16,8>, used at
src/sim_propagate.cpp:411,477) and src/detail/ival.hpp (used at :342-351).

Two legs: the committed golden vectors (tests/golden/ref_morton_ival.npz, produced by
tests/golden/make_golden.py from oracle/_ref) and, when oracle/_ref/libref_pin.so is
present (build container), a direct comparison on fresh random inputs.
"""
import ctypes as C
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_morton_ival.npz")


def test_morton_golden(orc):
    g = np.load(GOLD)
    L = orc.lib()
    got = np.array([L.orc_morton4(int(a), int(b), int(c), int(d)) for a, b, c, d in g["morton_fields"]], dtype=np.uint64)
    assert np.array_equal(got, g["morton_codes"])
    # x is the least significant lane of each nibble, r the most significant.
    assert L.orc_morton4(1, 0, 0, 0) == 0x1 and L.orc_morton4(0, 0, 0, 1) == 0x8


def test_ival_horner_golden(orc):
    g = np.load(GOLD)
    L = orc.lib()
    out = np.empty(2)
    for k in range(len(g["ival_cf"])):
        cf = np.ascontiguousarray(g["ival_cf"][k])
        L.orc_ival_horner(orc._p(cf), 20, float(g["ival_lo"][k]), float(g["ival_hi"][k]), orc._p(out))
        assert np.array_equal(out, g["ival_out"][k])  # bit-exact


def test_against_ref_library_if_present(orc):
    R = orc.ref_lib()
    if R is None:
        import pytest

        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    L = orc.lib()
    rng = np.random.default_rng(1)
    f = rng.integers(0, 65536, size=(2000, 4), dtype=np.uint64)
    for a, b, c, d in f:
        assert L.orc_morton4(int(a), int(b), int(c), int(d)) == R.ref_morton_encode(int(a), int(b), int(c), int(d))
    o1, o2 = np.empty(2), np.empty(2)
    for _ in range(200):
        cf = rng.standard_normal(21)
        lo = float(rng.uniform(0, 0.3))
        hi = lo + float(rng.uniform(0, 0.3))
        L.orc_ival_horner(orc._p(cf), 20, lo, hi, orc._p(o1))
        R.ref_ival_horner(cf.ctypes.data_as(C.c_void_p), 20, lo, hi, o2.ctypes.data_as(C.c_void_p))
        assert np.array_equal(o1, o2)
