"""jax.numpy stand-in: forwards to NumPy, results carry an `.at` property."""
import sys as _sys
import types as _types
import numpy as _np


class _At:
    def __init__(self, arr):
        self._arr = arr

    def __getitem__(self, idx):
        return _AtIdx(self._arr, idx)


def _fix_idx(idx):
    # jax accepts tuples of tuples as fancy indices; NumPy wants lists there
    if isinstance(idx, tuple):
        return tuple(list(i) if isinstance(i, tuple) else i for i in idx)
    return idx


class _AtIdx:
    def __init__(self, arr, idx):
        self._arr, self._idx = arr, _fix_idx(idx)

    def set(self, v):
        out = _np.array(self._arr, copy=True)
        out[self._idx] = v
        return _wrap(out)

    def add(self, v):
        out = _np.array(self._arr, copy=True)
        out[self._idx] += v
        return _wrap(out)

    def divide(self, v):
        out = _np.array(self._arr, copy=True)
        out[self._idx] /= v
        return _wrap(out)

    def get(self):
        return _wrap(_np.asarray(self._arr)[self._idx])


class ndarray(_np.ndarray):
    @property
    def at(self):
        return _At(self)

    def __getitem__(self, idx):
        return _wrap(_np.ndarray.__getitem__(_np.asarray(self), _fix_idx(idx)))


def _wrap(x):
    if isinstance(x, _np.ndarray):
        return x.view(ndarray)
    if isinstance(x, tuple):
        return tuple(_wrap(i) for i in x)
    return x


def array(x, dtype=None, **kw):
    return _wrap(_np.array(x, dtype=dtype))


def where(cond, *args, size=None, fill_value=0):
    if args:
        return _wrap(_np.where(cond, *args))
    idx = _np.nonzero(_np.asarray(cond))
    if size is not None:
        out = []
        for i in idx:
            pad = _np.full(size, fill_value, dtype=i.dtype)
            m = min(size, i.shape[0])
            pad[:m] = i[:m]
            out.append(_wrap(pad))
        return tuple(out)
    return tuple(_wrap(i) for i in idx)


int8 = _np.int8
float64 = _np.float64


class _Module(_types.ModuleType):
    def __getattr__(self, name):
        obj = getattr(_np, name)
        if callable(obj) and not isinstance(obj, type):
            def fwd(*a, **k):
                return _wrap(obj(*a, **k))
            fwd.__name__ = name
            return fwd
        return obj


_sys.modules[__name__].__class__ = _Module
