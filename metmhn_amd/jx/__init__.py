"""Mirrors of the metmhn.jx primitives (reference argument orders) on the GPU engine."""
from __future__ import annotations

import numpy as np

from ..engine import Engine

_ENG: dict = {}


def engine(n_mut: int, dtype: str = "f64") -> Engine:
    from ..engine import default_device
    key = (n_mut, dtype, default_device())
    if key not in _ENG:
        _ENG[key] = Engine(n_mut, dtype=dtype)
    return _ENG[key]


def n_from_joint(state) -> int:
    return (np.asarray(state).shape[0] - 1) // 2
