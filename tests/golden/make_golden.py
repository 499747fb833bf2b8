"""Generates tests/golden/ref_morton_ival.npz from oracle/_ref/libref_pin.so, i.e. from the
reference's own src/detail/mortonND_LUT.h and src/detail/ival.hpp compiled where they lie
(make -C oracle ref). Run in the build container (needs /root/reference); the .npz is
committed, the reference sources never are.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc_py  # noqa: E402

orc_py.build()
R = orc_py.ref_lib()
assert R is not None, "oracle/_ref/libref_pin.so missing: run `make -C oracle ref`"

rng = np.random.default_rng(20251110)
# Morton: random 16-bit fields + edge values.
f = rng.integers(0, 65536, size=(4096, 4), dtype=np.uint64)
edge = np.array(
    [[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [65535, 65535, 65535, 65535], [65535, 0, 0, 0], [0, 0, 0, 65535], [0x8000, 0x4000, 0x2000, 0x1000], [0xAAAA, 0x5555, 0xFFFF, 0x0001]],
    dtype=np.uint64,
)
f = np.concatenate([edge, f])
codes = np.array([R.ref_morton_encode(int(a), int(b), int(c), int(d)) for a, b, c, d in f], dtype=np.uint64)

# Interval Horner: random degree-20 polynomials on random sub-intervals of [0, 0.6].
P = 512
cf = rng.standard_normal((P, 21)) * np.exp(-0.4 * np.arange(21))[None, :]
lo = rng.uniform(0.0, 0.3, size=P)
hi = lo + rng.uniform(0.0, 0.3, size=P)
lo[:8] = 0.0
out = np.empty((P, 2))
tmp = np.empty(2)
for k in range(P):
    R.ref_ival_horner(cf[k].ctypes.data_as(C.c_void_p), 20, float(lo[k]), float(hi[k]), tmp.ctypes.data_as(C.c_void_p))
    out[k] = tmp

np.savez_compressed(os.path.join(os.path.dirname(__file__), "ref_morton_ival.npz"), morton_fields=f, morton_codes=codes, ival_cf=cf, ival_lo=lo, ival_hi=hi, ival_out=out)
print("wrote ref_morton_ival.npz:", f.shape, cf.shape)
