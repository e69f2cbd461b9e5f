"""metmhn/jx/one_event.py entry points: paired rows whose only active slot is the seeding (k = 1 joint spaces, which the
reference special-cases because `reshape(-1, 4)` cannot trace there, one_event.py:5-7).  The engine has no special case: a k = 1
space is a one-tile problem like any other; these mirrors exist so that callers of the reference's names find them
(regularized_optimization.py:103-104 dispatches here when n_prim + n_met - 1 == 1)."""
from __future__ import annotations

from .likelihood import (_lp_coupled_0, _lp_coupled_1, _lp_coupled_2,          # noqa: F401  (one_event.py:141-228: same
                         _g_coupled_0, _g_coupled_1, _g_coupled_2)             #  arguments minus n_prim / n_met; :307-409)
