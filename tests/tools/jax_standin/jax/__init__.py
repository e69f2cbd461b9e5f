"""Eager NumPy stand-in for the small `jax` subset the metMHN hot path uses.

TEST TOOLING ONLY.  It exists so that the reference's *own Python source*
(/root/reference, never copied) can be executed in the build container, where
jax/jaxlib are not installed, to (1) check the CPU oracle under oracle/ and
(2) emit the golden vectors committed under tests/golden/ (see
tests/tools/make_golden.py).  Nothing on the product path imports this.

Semantics: every jnp call is the NumPy call of the same name evaluated eagerly
in float64; `.at[idx].set/add/divide/get` are copy-on-write; `jit` is the
identity; `vmap` is a Python loop + stack; `lax` control flow runs in Python.
"""
import numpy as _np
from . import numpy  # noqa: F401  (jax.numpy)
from . import lax    # noqa: F401
from . import random  # noqa: F401


class _Config:
    def update(self, *a, **k):
        return None


config = _Config()


def jit(fun=None, **kw):
    if fun is None:
        return lambda f: f
    return fun


def vmap(fun, in_axes=0, out_axes=0):
    def wrapped(*args):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        size = None
        for a, ax in zip(args, axes):
            if ax is not None:
                size = _np.asarray(a).shape[ax]
                break
        outs = []
        for t in range(size):
            call = [a if ax is None else numpy._wrap(_np.take(_np.asarray(a), t, axis=ax))
                    for a, ax in zip(args, axes)]
            outs.append(fun(*call))
        if isinstance(outs[0], tuple):
            return tuple(numpy._wrap(_np.stack([_np.asarray(o[i]) for o in outs], axis=0))
                         for i in range(len(outs[0])))
        return numpy._wrap(_np.stack([_np.asarray(o) for o in outs], axis=0))
    return wrapped
