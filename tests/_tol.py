"""Tolerance of BASELINE.json's contract for propagated states: 1e-9 RELATIVE, taken at the scale of
the vector a coordinate belongs to (|r| for x, y, z; |v| for vx, vy, vz) -- a coordinate that happens
to pass through zero has no scale of its own. The one-lane-per-object kernels repeat the oracle's
operations in the oracle's order (bit-identical in practice); k_propagate_wave adds its convolution
sums in another order and differs by rounding (1e-16 per superstep, profiles/exp_wave_accuracy.py)."""
import numpy as np


def states_close(a, b, rtol=1e-9):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    ok = True
    for sl in (slice(0, 3), slice(3, 6)):
        scale = np.maximum(1.0, np.linalg.norm(b[:, sl], axis=1))
        ok = ok and bool(np.all(np.linalg.norm(a[:, sl] - b[:, sl], axis=1) <= rtol * scale))
    if a.shape[1] > 6:
        ok = ok and np.array_equal(a[:, 6:], b[:, 6:])
    return ok
