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
