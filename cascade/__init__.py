"""``import cascade``: the reference's package name (cascade.py/__init__.py:47-61) over the
MI355X-native implementation in ``cascade_amd`` -- ``cascade.sim``, ``cascade.outcome``,
``cascade.dynamics.kepler`` / ``simple_earth`` / ``mascon_asteroid`` / ``mascon_asteroid_energy``,
``cascade.set_nthreads`` / ``get_nthreads`` and the six ``cascade.set_logger_level_*``.
``cascade.test`` is not part of this build."""
import sys as _sys

import cascade_amd as _impl
from cascade_amd import dynamics, outcome, sim  # noqa: F401
from cascade_amd.core import (  # noqa: F401
    get_nthreads,
    set_logger_level_critical,
    set_logger_level_debug,
    set_logger_level_err,
    set_logger_level_info,
    set_logger_level_trace,
    set_logger_level_warn,
    set_nthreads,
)

core = _impl.core
_sys.modules[__name__ + ".dynamics"] = dynamics
_sys.modules[__name__ + ".core"] = core
__all__ = ["sim", "outcome", "dynamics", "core", "set_nthreads", "get_nthreads"] + ["set_logger_level_" + k for k in ("trace", "debug", "info", "warn", "err", "critical")]
