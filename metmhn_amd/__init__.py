"""metmhn_amd - MI355X-native likelihood / gradient engine for metMHN.

Drop-in for the hot path of cbg-ethz/metMHN (metmhn.regularized_optimization and the
metmhn.jx primitives it calls): same function names and argument orders, computed by
hand-written HIP kernels behind the C ABI of include/metmhn_amd.h.  `model.MetMHN` / `state`
mirror metmhn.model / metmhn.state (order likelihoods and likeliest orders) on top of it.
"""
from . import _lib  # noqa: F401
from .engine import Engine  # noqa: F401

__all__ = ["Engine", "regularized_optimization", "distributed", "synthetic", "model", "state"]
