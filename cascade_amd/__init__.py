"""cascade_amd: MI355X-native implementation of cascade's per-superstep pipeline
(cascade::sim::step) behind the reference's ``cascade.sim`` interface."""
from . import dynamics  # noqa: F401
from .sim import outcome, sim  # noqa: F401

__all__ = ["sim", "outcome", "dynamics"]
