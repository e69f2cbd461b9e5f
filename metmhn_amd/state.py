"""Event sets as bit masks: the host-side mirror of metmhn/state.py (State, MetState).

Only what the order-inference path (model.py) and its callers need: a set of event indices stored
as one Python int, convertible from / to the 0/1 vectors the rest of the package uses.  A
`MetState` over n events has 2n+1 slots laid out as the reference lays them out (state.py:201-300):
slot 2i = event i in the primary tumour, slot 2i+1 = event i in the metastasis, slot 2n = seeding.
"""
from __future__ import annotations

from collections.abc import Iterable, MutableSet

import numpy as np


class _Bits(MutableSet):
    """A mutable set of small non-negative ints backed by one integer mask."""

    __slots__ = ("_mask", "_size")

    def __init__(self, data, /, size: int):
        if isinstance(data, (int, np.integer)) and not isinstance(data, bool):
            if data < 0:
                raise ValueError("The given integer must be non-negative")
            self._mask = int(data)
        elif isinstance(data, Iterable):
            self._mask = 0
            for item in data:
                self._mask |= 1 << int(item)
        else:
            raise TypeError(f"unsupported argument type: '{type(data).__name__}'")
        self._size = int(size)

    data = property(lambda self: self._mask)
    size = property(lambda self: self._size)

    def __contains__(self, item) -> bool:
        return bool(self._mask >> int(item) & 1)

    def __iter__(self):
        mask, pos = self._mask, 0
        while mask:
            if mask & 1:
                yield pos
            mask >>= 1
            pos += 1

    def __len__(self) -> int:
        return bin(self._mask).count("1")

    def add(self, item) -> None:
        self._mask |= 1 << int(item)

    def discard(self, item) -> None:
        self._mask &= ~(1 << int(item))

    def __hash__(self) -> int:
        return hash((type(self).__name__, self._mask, self._size))

    def __eq__(self, other) -> bool:
        return isinstance(other, _Bits) and (self._mask, self._size) == (other._mask, other._size)

    def __repr__(self) -> str:
        return f"{type(self).__name__}({list(self)}, size={self._size})"

    @classmethod
    def from_seq(cls, seq, /, labels=None):
        """From a 0/1 vector (state.py:93-96, 336-339)."""
        return cls((i for i, bit in enumerate(seq) if bit), size=len(seq))

    def to_seq(self) -> np.ndarray:
        seq = np.zeros(self._size, dtype=bool)
        seq[list(self)] = True
        return seq


class State(_Bits):
    """Events of ONE tumour (state.py:16-101)."""
    __slots__ = ()


class MetState(_Bits):
    """Joint primary-tumour / metastasis observation (state.py:201-345)."""
    __slots__ = ()

    @property
    def n(self) -> int:
        return self._size // 2

    @property
    def events(self) -> tuple:
        return tuple(self)

    @property
    def PT_events(self) -> tuple:
        return tuple(i for i in range(self.n) if self._mask >> (2 * i) & 1)

    @property
    def MT_events(self) -> tuple:
        return tuple(i for i in range(self.n) if self._mask >> (2 * i + 1) & 1)

    @property
    def Seeding(self) -> tuple:
        return (self.n,) if self._mask >> (self._size - 1) & 1 else ()

    @property
    def PT(self) -> State:
        return State(self.PT_events, size=self.n)

    @property
    def PT_S(self) -> State:
        return State(self.PT_events + self.Seeding, size=self.n + 1)

    @property
    def MT(self) -> State:
        """Metastasis events plus the seeding; EMPTY when the seeding is absent (state.py:277-281)."""
        return State(self.MT_events + self.Seeding if self.Seeding else (), size=self.n + 1)

    @property
    def reachable(self) -> bool:
        """Before the seeding both tumours carry the same events (state.py:283-287)."""
        return bool(self.Seeding) or self.PT_events == self.MT_events
