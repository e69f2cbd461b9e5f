"""Patient sharding across GPUs and the one collective of an evaluation.

Patients are independent given (theta, d_p, d_m) and the objective is a weighted sum
(regularized_optimization.py:256-266), so every rank evaluates its own shard and ONE
all-reduce of the unweighted partial sums (4 + 2 N^2 + 3 N doubles, ~7.5 KB at n = 20)
per evaluation combines them; the EM/NM weight w only needs the global counts, which
travel in the same buffer.  One process per GPU; torch.distributed supplies the
collective ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def patient_cost(dat: np.ndarray) -> np.ndarray:
    """Work estimate per row: 2^k (k+1) state updates per solve."""
    dat = np.asarray(dat)
    typ = dat[:, -1]
    bits = dat[:, :-2].astype(np.int64)
    k_joint = bits.sum(axis=1)
    k_pt = bits[:, 0::2].sum(axis=1)
    k_mt = bits[:, 1:-1:2].sum(axis=1) + 1
    k = np.where(typ == 3, k_joint, np.where(typ == 2, k_mt, k_pt))
    return np.exp2(k.astype(np.float64)) * (k + 1)


def shard_rows(dat: np.ndarray, world_size: int) -> list[np.ndarray]:
    """Static longest-processing-time partition of the rows; deterministic."""
    cost = patient_cost(dat)
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world_size)
    parts: list[list[int]] = [[] for _ in range(world_size)]
    for r in order:
        w = int(np.argmin(load))
        parts[w].append(int(r))
        load[w] += cost[r]
    return [np.sort(np.array(p, dtype=np.int64)) for p in parts]


def fold_jobs(n_lambda: int, n_folds: int, rank: int = 0, world_size: int = 1) -> list[tuple[int, int]]:
    """(penalty index, fold) jobs of cross_val owned by `rank`: the reference's loop order (Utilityfunctions.py:
    207-208), dealt round-robin so that every rank gets the same number of fits (+-1)."""
    jobs = [(i, f) for i in range(n_lambda) for f in range(n_folds)]
    return jobs[rank::world_size]


class CommAbandoned(RuntimeError):
    """A rank gave up on the collective set-up of the in-library communicator (another rank failed, or it timed out)."""


_JOIN_GEN = [0]


def collective_init(init_fn, rank: int, world: int, store, timeout: float = 180.0, poll: float = 0.2) -> tuple[bool, str]:
    """Run `init_fn()` (a blocking collective set-up: ncclCommInitRank) on every rank and decide TOGETHER whether it
    succeeded, without ever leaving a rank alone in it for long.  Returns (all_ok, first error message).

    Protocol over the torch.distributed key-value store (no collective: a collective would hang on the ranks that are
    still blocked inside init_fn).  Every call has a generation number (all ranks call in the same order); a rank writes
    ONE status key  mmhn_comm/<gen>/<rank> = "ok" | "err:<message>" | "abandoned:<why>":
      * init_fn returns            -> "ok"
      * init_fn raises             -> "err:..."  (a rank that fails BEFORE entering the set-up - the library is missing, a
                                      device is gone: the case in which its peers would sit in ncclCommInitRank until the
                                      timeout - is what the others see within `poll` seconds)
      * init_fn is still blocked while another rank's key says err / abandoned, or after `timeout` seconds
                                   -> "abandoned:...", and CommAbandoned is raised: the blocked thread cannot be
                                      cancelled, so this process is expected to end (the job dies in seconds, not after
                                      MMHN_COMM_TIMEOUT).
    Once every rank's key is there: all "ok" -> (True, ""); an "abandoned" anywhere -> CommAbandoned on every rank (a
    process is about to die: no fallback can be collective); otherwise (errors only, nobody blocked) -> (False, message):
    the caller's ranks fall back together."""
    import threading
    import time
    gen = _JOIN_GEN[0]
    _JOIN_GEN[0] += 1
    res = {}

    def _run():
        try:
            init_fn()
            res["ok"] = True
        except Exception as exc:                               # noqa: BLE001
            res["err"] = f"{type(exc).__name__}: {exc}"

    th = threading.Thread(target=_run, daemon=True)
    th.start()
    if world == 1 or store is None:
        th.join(timeout)
        if th.is_alive():
            raise CommAbandoned("communicator set-up did not return within the timeout")
        return bool(res.get("ok")), res.get("err", "")
    key = lambda r: f"mmhn_comm/{gen}/{r}"                    # noqa: E731

    def peers():
        out = {}
        for r in range(world):
            if r != rank and store.check([key(r)]):
                out[r] = store.get(key(r)).decode()
        return out

    t0 = time.monotonic()
    while True:
        th.join(poll)
        if not th.is_alive():
            break
        bad = {r: v for r, v in peers().items() if not v.startswith("ok")}
        if bad:
            r, v = next(iter(bad.items()))
            store.set(key(rank), f"abandoned:rank {r} reported {v}")
            raise CommAbandoned(f"rank {r} failed while this rank was still inside the communicator set-up ({v}); giving up")
        if time.monotonic() - t0 > timeout:
            store.set(key(rank), "abandoned:timeout")
            raise CommAbandoned(f"communicator set-up did not return within {timeout:.0f} s - a rank is missing")
    mine = "ok" if res.get("ok") else "err:" + res.get("err", "unknown")
    store.set(key(rank), mine)
    while True:
        got = peers()
        if len(got) == world - 1:
            break
        if any(v.startswith("abandoned") for v in got.values()) or time.monotonic() - t0 > timeout:
            break
        time.sleep(poll)
    got[rank] = mine
    gone = {r: v for r, v in got.items() if v.startswith("abandoned")}
    if gone or len(got) < world:
        raise CommAbandoned(f"communicator set-up abandoned by {sorted(gone) or 'a rank that never reported'}: {gone}")
    errs = [f"rank {r}: {v[4:]}" for r, v in sorted(got.items()) if v.startswith("err:")]
    return (not errs), "; ".join(errs)


def allreduce_sums(sums: np.ndarray, group=None) -> np.ndarray:
    """Sum the partial-sum buffers of all ranks (no-op without an initialised process group)."""
    import torch
    import torch.distributed as dist
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return sums
    if dist.get_world_size(group) == 1 and os.environ.get("MMHN_FORCE_ALLREDUCE") != "1":
        return sums
    t = torch.from_numpy(np.ascontiguousarray(sums, dtype=np.float64))
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()


def allreduce_sums_fixed_order(sums: np.ndarray, group=None) -> np.ndarray:
    """MMHN_REDUCE=host_fixed_order: every rank gets every rank's partial buffer and adds them up in rank order on
    the host - the same association on every rank and in every run (reproducible across runs of one GPU
    count; the in-library RCCL all-reduce leaves the association to the library)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sums
    t = torch.from_numpy(np.ascontiguousarray(sums, dtype=np.float64))
    on_gpu = dist.get_backend(group) == "nccl"
    if on_gpu:
        t = t.cuda()
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t, group=group)
    out = np.zeros_like(np.asarray(sums, dtype=np.float64))
    for part in parts:                                        # rank 0 first, always
        out = out + part.cpu().numpy()
    return out


def em_weight(n_em: float, n_pat: float, perc_met: float):
    """(w, n_full) of regularized_optimization.py:121-128 from the GLOBAL counts.  perc_met = 1 with both kinds of
    rows present makes the reference's weight infinite (its score NaN): rejected here instead of reaching the device."""
    n_nm = n_pat - n_em
    if n_em * n_nm != 0 and not (0.0 <= float(perc_met) < 1.0):
        raise ValueError(f"perc_met must be in [0, 1) for a cohort with EM and NM rows, got {perc_met!r}")
    w = perc_met * n_nm / ((1 - perc_met) * n_em) if n_em * n_nm != 0 else 1.0
    return w, w * n_em + n_nm


def split_wsums(ws: np.ndarray, N: int, n_full: float):
    """(score, d_theta, d_dp, d_dm) from the pre-combined buffer of mmhn_cohort_wsums."""
    o = 1
    g = ws[o:o + N * N].reshape(N, N); o += N * N
    p = ws[o:o + N]; o += N
    m = ws[o:o + N]
    return ws[0] / n_full, g / n_full, p / n_full, m / n_full


def combine_sums(sums: np.ndarray, N: int, perc_met: float):
    """(score, d_theta, d_dp, d_dm) from the buffer of mmhn_cohort_sums
    (regularized_optimization.py:256-266)."""
    s_em, s_nm, n_em, n_pat = sums[:4]
    n_nm = n_pat - n_em
    w = perc_met * n_nm / ((1 - perc_met) * n_em) if n_em * n_nm != 0 else 1.0
    n_full = w * n_em + n_nm
    o = 4
    g_em = sums[o:o + N * N].reshape(N, N); o += N * N
    g_nm = sums[o:o + N * N].reshape(N, N); o += N * N
    p_em = sums[o:o + N]; o += N
    p_nm = sums[o:o + N]; o += N
    m_em = sums[o:o + N]
    return ((w * s_em + s_nm) / n_full, (w * g_em + g_nm) / n_full, (w * p_em + p_nm) / n_full,
            w * m_em / n_full)
