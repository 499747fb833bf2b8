"""cascade_amd: MI355X-native implementation of cascade's per-superstep pipeline
(cascade::sim::step) behind the reference's ``cascade.sim`` interface.

``sim`` and ``outcome`` come from the pybind11 module ``cascade_amd.core`` (the counterpart of
the reference's ``cascade.core``, cascade.py/core.cpp), which wraps the C++ host class
``cascade::sim`` (include/cascade/sim.hpp, libcascade.so) that drives the gfx950 kernels of
libcascade_hip.so through the C ABI of include/cascade_hip.h. There is no pure-Python or CPU
implementation behind these names: a missing extension is an ImportError."""
try:
    from .core import outcome, sim  # noqa: F401
except ImportError as e:  # pragma: no cover
    raise ImportError("cascade_amd.core is not built: run `python -c 'import __graft_entry__ as g; g.build()'` in the repository root (" + str(e) + ")") from e
from . import dynamics  # noqa: F401,E402

__all__ = ["sim", "outcome", "dynamics"]
