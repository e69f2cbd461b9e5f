"""jax.lax stand-in: Python control flow."""
import numpy as _np
from . import numpy as _jnp


def fori_loop(lower, upper, body_fun, init_val):
    val = init_val
    for i in range(int(lower), int(upper)):
        val = body_fun(i, val)
    return val


def while_loop(cond_fun, body_fun, init_val):
    val = init_val
    while cond_fun(val):
        val = body_fun(val)
    return val


def switch(index, branches, *operands, operand=None):
    i = int(_np.clip(int(index), 0, len(branches) - 1))
    if operand is not None and not operands:
        return branches[i](operand)
    return branches[i](*operands)


def cond(pred, true_fun, false_fun, *operands):
    return true_fun(*operands) if bool(pred) else false_fun(*operands)


def select_n(which, *cases):
    return cases[int(which)]


def select(pred, on_true, on_false):
    return _jnp._wrap(_np.where(pred, on_true, on_false))


def dynamic_slice(operand, start_indices, slice_sizes):
    sl = tuple(slice(int(s), int(s) + int(n)) for s, n in zip(start_indices, slice_sizes))
    return _jnp._wrap(_np.asarray(operand)[sl])
