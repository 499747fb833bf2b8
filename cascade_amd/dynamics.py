"""Built-in dynamics descriptors.

The reference passes dynamics as heyoka expressions (``kw::dyn``, include/cascade/sim.hpp:213-222;
``cascade.dynamics.kepler``, src/sim_dynamics.cpp:25-42). heyoka's JIT is replaced here by
hand-written Taylor kernels, so ``dyn`` is a small descriptor object (``cascade::dynamics::dyn_t``)
instead; the call signatures of the factory functions are the reference's.
"""
import math

from . import core


def kepler(mu: float = 1.0):
    """Keplerian dynamics (src/sim_dynamics.cpp:25-42)."""
    if not math.isfinite(mu) or mu <= 0:
        raise ValueError("The mu parameter for Keplerian dynamics must be finite and positive, but it is {} instead".format(mu))
    return core._kepler(float(mu))
