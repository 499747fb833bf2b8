"""Dynamics for ``cascade_amd.sim``.

The reference passes dynamics as heyoka expressions (``kw::dyn``, include/cascade/sim.hpp:213-222;
``cascade.dynamics.kepler``, src/sim_dynamics.cpp:25-42) and JIT-compiles an integrator for them.
Here the equations are ``cascade_amd.dynamics.expression`` objects (the C++ class
``cascade::expression``, include/cascade/expression.hpp): same way of writing them --
``make_vars``, ``par[i]``, ``time``, ``sin``/``cos``/``exp``/``sqrt``/``pow``/``diff`` -- but lowered
to a tape of Taylor-series operations that the GPU propagation kernel interprets. ``kepler()``
and ``simple_earth()`` have the reference's signatures and return a list-like object of six
``(variable, expression)`` pairs that can be modified item by item exactly like the reference's
lists (``dyn[3] = (dyn[3][0], dyn[3][1] + f)``); untouched Kepler / J2 / J2 + drag dynamics run on
hand-unrolled kernels.
"""
import math

from . import core
from .core import expression, lower_dynamics  # noqa: F401


def make_vars(*names):
    """Variables by name (heyoka.make_vars); a single name returns a single expression."""
    vs = [expression(str(n)) for n in names]
    return vs[0] if len(vs) == 1 else vs


class _Par:
    """``par[i]``: runtime parameter ``i`` of each object (heyoka.par)."""

    def __getitem__(self, i):
        return core._ex_par(int(i))


par = _Par()
#: the simulation time (heyoka.time)
time = core._ex_time()


def _ex(a):
    return a if isinstance(a, expression) else expression(float(a))


def sin(a):
    return core._ex_sin(_ex(a))


def cos(a):
    return core._ex_cos(_ex(a))


def exp(a):
    return core._ex_exp(_ex(a))


def sqrt(a):
    return core._ex_sqrt(_ex(a))


def pow(a, e):  # noqa: A001
    return core._ex_pow(_ex(a), float(e))


def diff(e, var):
    """Symbolic derivative of ``e`` with respect to the variable ``var`` (heyoka.diff)."""
    return core._ex_diff(_ex(e), var)


def kepler(mu: float = 1.0):
    """Keplerian dynamics (src/sim_dynamics.cpp:25-42)."""
    if not math.isfinite(mu) or mu <= 0:
        raise ValueError("The mu parameter for Keplerian dynamics must be finite and positive, but it is {} instead".format(mu))
    return core._kepler(float(mu))


def simple_earth(J2=True, J3=False, J4=False, C22S22=True, sun=False, moon=False, SRP=False, drag=True, thermonets=False):
    """Perturbed dynamics around the Earth in SI units: signature, defaults and formulas of the
    reference's ``cascade.dynamics.simple_earth`` (cascade.py/dynamics/_simple_earth.py:86-411).
    With ``drag`` every object's BSTAR is its first simulation parameter (``pars``), with ``SRP``
    the term Cr*A/m follows it. ``thermonets`` (a neural density model evaluated by heyoka) is not
    available and raises ``ValueError``."""
    return core._simple_earth(J2, J3, J4, C22S22, sun, moon, SRP, drag, thermonets)


def mascon_asteroid(Gconst, points, masses, omega):
    """Dynamics around a rotating irregular body made of point masses, written in the body-fixed
    rotating frame: signature and equations of the reference's ``cascade.dynamics.mascon_asteroid``
    (cascade.py/dynamics/_mascon_asteroid.py:13-41, which hands them to heyoka's mascon model):

        r' = v,   v' = -G sum_j m_j (r - r_j) / |r - r_j|^3 - 2 w x v - w x (w x r)

    ``points`` [N, 3], ``masses`` [N], ``omega`` [3], all in the caller's units. Returns six
    (variable, expression) pairs for ``sim(dyn=...)``; the equations run from their tape (one
    power, three products and four sums per mascon)."""
    import numpy as np

    pts = np.asarray(points, dtype=float)
    ms = np.asarray(masses, dtype=float).reshape(-1)
    om = np.asarray(omega, dtype=float).reshape(-1)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError("The mascon points must be an array of shape (N, 3), but the shape is {} instead".format(pts.shape))
    if ms.shape[0] != pts.shape[0]:
        raise ValueError("The number of masses ({}) differs from the number of mascon points ({})".format(ms.shape[0], pts.shape[0]))
    if om.shape[0] != 3:
        raise ValueError("The angular velocity must have 3 components, but it has {} instead".format(om.shape[0]))
    if not (np.isfinite(pts).all() and np.isfinite(ms).all() and np.isfinite(om).all() and math.isfinite(Gconst)):
        raise ValueError("The mascon model needs finite points, masses, angular velocity and gravitational constant")
    x, y, z, vx, vy, vz = make_vars("x", "y", "z", "vx", "vy", "vz")
    p, q, r = (float(v) for v in om)
    acc = [expression(0.0), expression(0.0), expression(0.0)]
    for (px, py, pz), m in zip(pts, ms):
        d = (x - float(px), y - float(py), z - float(pz))
        k = (-float(Gconst) * float(m)) * pow(d[0] * d[0] + d[1] * d[1] + d[2] * d[2], -1.5)
        acc = [a + k * dk for a, dk in zip(acc, d)]
    # w x v and w x (w x r)
    wv = (q * vz - r * vy, r * vx - p * vz, p * vy - q * vx)
    wr = (q * z - r * y, r * x - p * z, p * y - q * x)
    wwr = (q * wr[2] - r * wr[1], r * wr[0] - p * wr[2], p * wr[1] - q * wr[0])
    rhs = [a - 2.0 * c - cc for a, c, cc in zip(acc, wv, wwr)]
    return core.dyn_t([(x, vx), (y, vy), (z, vz), (vx, rhs[0]), (vy, rhs[1]), (vz, rhs[2])])


def mascon_asteroid_energy(state, Gconst, points, masses, omega):
    """The conserved energy (Jacobi integral) of a state ``[x, y, z, vx, vy, vz]`` in the dynamics
    of ``mascon_asteroid`` (cascade.py/dynamics/_mascon_asteroid.py:43-68): kinetic + gravitational
    + centrifugal potential."""
    import numpy as np

    s = np.asarray(state, dtype=float)
    pts = np.asarray(points, dtype=float)
    ms = np.asarray(masses, dtype=float).reshape(-1)
    om = np.asarray(omega, dtype=float).reshape(-1)
    pos, vel = s[..., 0:3], s[..., 3:6]
    kinetic = 0.5 * np.sum(vel * vel, axis=-1)
    dist = np.sqrt(np.sum((pos[..., None, :] - pts) ** 2, axis=-1))
    grav = -float(Gconst) * np.sum(ms / dist, axis=-1)
    centrifugal = -0.5 * np.dot(om, om) * np.sum(pos * pos, axis=-1) + 0.5 * np.sum(pos * om, axis=-1) ** 2
    return kinetic + grav + centrifugal
