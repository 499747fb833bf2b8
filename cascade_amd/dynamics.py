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


def simple_earth(J2=True, J3=False, J4=False, C22S22=True, sun=False, moon=False, SRP=False, drag=True, thermonets=False):
    """Perturbed dynamics around the Earth in SI units: signature and defaults of the reference's
    ``cascade.dynamics.simple_earth`` (cascade.py/dynamics/_simple_earth.py:86-96). Built so
    far: ``J2`` and ``drag`` (exponential density fit; pass every object's BSTAR as its first
    simulation parameter, ``pars``). Any other term raises ``ValueError`` -- note that the
    reference switches ``C22S22`` on by default, so it has to be turned off explicitly here."""
    return core._simple_earth(J2, J3, J4, C22S22, sun, moon, SRP, drag, thermonets)
