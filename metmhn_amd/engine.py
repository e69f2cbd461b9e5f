"""Python handle on the native engine (one engine = one GPU)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import f64p, i8p, i64p

F64, F32 = 0, 1
_DEFAULT = {"device": None}


def set_default_device(device: int | None):
    """GPU used by every Engine created without an explicit `device` (jx mirrors, simulations, the objective).
    None: LOCAL_RANK of a torch.distributed launch, else 0."""
    _DEFAULT["device"] = None if device is None else int(device)


def default_device() -> int:
    if _DEFAULT["device"] is not None:
        return _DEFAULT["device"]
    import os
    return int(os.environ.get("LOCAL_RANK", "0"))


def unique_id() -> bytes:
    """128-byte RCCL id for Engine.comm_init (call on one rank, hand to the others by any host channel)."""
    buf = C.create_string_buffer(128)
    _lib.check(_lib.load().mmhn_comm_unique_id(buf))
    return buf.raw


def _f(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a, a.ctypes.data_as(f64p)


def _s(a):
    a = np.ascontiguousarray(np.asarray(a).astype(np.int8))
    return a, a.ctypes.data_as(i8p)


class Engine:
    """Owns a mmhn_handle.  `n_mut` mutations -> N = n_mut + 1 events incl. seeding."""

    def __init__(self, n_mut: int, device: int | None = None, dtype: str = "f64", workspace_bytes: int | None = None):
        self.lib = _lib.load()
        if device is None:
            device = default_device()
        self.device = int(device)
        self.n = int(n_mut)
        self.N = self.n + 1
        self.dtype = dtype
        h = C.c_void_p()
        _lib.check(self.lib.mmhn_create(int(device), self.n, F64 if dtype == "f64" else F32, C.byref(h)))
        self.h = h
        self.n_pat = 0
        self.comm_size = 1
        if workspace_bytes:
            _lib.check(self.lib.mmhn_set_workspace_limit(self.h, int(workspace_bytes)))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if getattr(self, "h", None):
            self.lib.mmhn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- cohort objective
    def set_cohort(self, dat):
        dat = np.ascontiguousarray(np.asarray(dat).astype(np.int8))
        if dat.ndim != 2 or dat.shape[1] != 2 * self.n + 3:
            raise ValueError(f"dat must be [n_pat, {2 * self.n + 3}]")
        _lib.check(self.lib.mmhn_set_cohort(self.h, dat.ctypes.data_as(i8p), dat.shape[0], dat.shape[1]))
        self.n_pat = dat.shape[0]

    def _params(self, log_theta, log_d_p, log_d_m):
        lt, ltp = _f(log_theta)
        dp, dpp = _f(log_d_p)
        dm, dmp = _f(log_d_m)
        if lt.shape != (self.N, self.N) or dp.shape != (self.N,) or dm.shape != (self.N,):
            raise ValueError("parameter shapes do not match n_mut")
        return (lt, dp, dm), (ltp, dpp, dmp)

    def score(self, log_theta, log_d_p, log_d_m, perc_met):
        keep, (a, b, c) = self._params(log_theta, log_d_p, log_d_m)
        out = C.c_double()
        _lib.check(self.lib.mmhn_score(self.h, a, b, c, float(perc_met), C.byref(out)))
        return out.value

    def score_and_grad(self, log_theta, log_d_p, log_d_m, perc_met):
        keep, (a, b, c) = self._params(log_theta, log_d_p, log_d_m)
        out = C.c_double()
        g = np.zeros((self.N, self.N))
        gp = np.zeros(self.N)
        gm = np.zeros(self.N)
        _lib.check(self.lib.mmhn_score_and_grad(self.h, a, b, c, float(perc_met), C.byref(out),
                                                 g.ctypes.data_as(f64p), gp.ctypes.data_as(f64p),
                                                 gm.ctypes.data_as(f64p)))
        return out.value, g, gp, gm

    def cohort_sums(self, log_theta, log_d_p, log_d_m, with_grad=True):
        keep, (a, b, c) = self._params(log_theta, log_d_p, log_d_m)
        sums = np.zeros(4 + 2 * self.N * self.N + 3 * self.N)
        _lib.check(self.lib.mmhn_cohort_sums(self.h, a, b, c, int(bool(with_grad)), sums.ctypes.data_as(f64p)))
        return sums

    def cohort_sums_begin(self, log_theta, log_d_p, log_d_m, with_grad=True):
        """Issue the evaluation and return at once (mmhn_cohort_sums_begin); collect with cohort_sums_end()."""
        keep, (a, b, c) = self._params(log_theta, log_d_p, log_d_m)
        _lib.check(self.lib.mmhn_cohort_sums_begin(self.h, a, b, c, int(bool(with_grad))))

    def cohort_sums_end(self):
        sums = np.zeros(4 + 2 * self.N * self.N + 3 * self.N)
        _lib.check(self.lib.mmhn_cohort_sums_end(self.h, sums.ctypes.data_as(f64p)))
        return sums

    def cohort_wsums_begin(self, log_theta, log_d_p, log_d_m, w, with_grad=True, flag=None):
        """Like cohort_sums_begin with the EM / NM weighting applied on the device (mmhn_cohort_wsums_begin).
        flag (optional): one more double that rides in the same all-reduce (mmhn_set_reduce_flag); its sum over the ranks is
        `reduce_flag` after cohort_wsums_end."""
        keep, (a, b, c) = self._params(log_theta, log_d_p, log_d_m)
        if flag is not None:
            _lib.check(self.lib.mmhn_set_reduce_flag(self.h, float(flag)))
        _lib.check(self.lib.mmhn_cohort_wsums_begin(self.h, a, b, c, int(bool(with_grad)), float(w)))

    def cohort_wsums_end(self):
        ws = np.zeros(1 + self.N * self.N + 2 * self.N)
        _lib.check(self.lib.mmhn_cohort_wsums_end(self.h, ws.ctypes.data_as(f64p)))
        out = C.c_double()
        _lib.check(self.lib.mmhn_get_reduce_flag(self.h, C.byref(out)))
        self.reduce_flag = out.value
        return ws

    def patient_grads(self, log_theta, log_d_p, log_d_m, with_grad=True):
        keep, (a, b, c) = self._params(log_theta, log_d_p, log_d_m)
        P, N = self.n_pat, self.N
        lp = np.zeros(P)
        if not with_grad:
            _lib.check(self.lib.mmhn_patient_grads(self.h, a, b, c, lp.ctypes.data_as(f64p), None, None, None))
            return lp
        g, gp, gm = np.zeros((P, N, N)), np.zeros((P, N)), np.zeros((P, N))
        _lib.check(self.lib.mmhn_patient_grads(self.h, a, b, c, lp.ctypes.data_as(f64p), g.ctypes.data_as(f64p),
                                               gp.ctypes.data_as(f64p), gm.ctypes.data_as(f64p)))
        return lp, g, gp, gm

    # ---- joint primitives
    def kronvec(self, log_theta, p, state, diag=True, transpose=False):
        lt, ltp = _f(log_theta); pv, pp = _f(p); st, sp = _s(state)
        y = np.zeros_like(pv)
        _lib.check(self.lib.mmhn_kronvec(self.h, ltp, sp, pp, y.ctypes.data_as(f64p), int(diag), int(transpose)))
        return y

    def kronvec_batched(self, log_theta, p, state, diag=True, transpose=False):
        """kronvec for a batch p[b][2^k] of vectors of one restricted space: one launch, y[b][2^k]."""
        lt, ltp = _f(log_theta); st, sp = _s(state)
        pv = np.ascontiguousarray(p, dtype=np.float64)
        if pv.ndim != 2 or pv.shape[1] != 2 ** int(st.sum()):
            raise ValueError("p must have shape [batch, 2^k]")
        y = np.zeros_like(pv)
        _lib.check(self.lib.mmhn_kronvec_batched(self.h, ltp, sp, int(pv.shape[0]), pv.ctypes.data_as(f64p),
                                                 y.ctypes.data_as(f64p), int(diag), int(transpose)))
        return y

    def jacobi_step_batched(self, log_theta, log_d_p, log_d_m, p, rhs, state, transpose=False):
        """One sweep of R_i_inv_vec's iteration for a batch: lidg * (Q_off p + rhs), shapes [batch, 2^k]."""
        lt, ltp = _f(log_theta); a, ap = _f(log_d_p); b, bp = _f(log_d_m); st, sp = _s(state)
        pv = np.ascontiguousarray(p, dtype=np.float64)
        rv = np.ascontiguousarray(rhs, dtype=np.float64)
        if pv.ndim != 2 or pv.shape != rv.shape or pv.shape[1] != 2 ** int(st.sum()):
            raise ValueError("p and rhs must have shape [batch, 2^k]")
        y = np.zeros_like(pv)
        _lib.check(self.lib.mmhn_jacobi_step_batched(self.h, ltp, ap, bp, sp, int(pv.shape[0]), pv.ctypes.data_as(f64p),
                                                     rv.ctypes.data_as(f64p), y.ctypes.data_as(f64p), int(transpose)))
        return y

    def kron_diag(self, log_theta, state):
        lt, ltp = _f(log_theta); st, sp = _s(state)
        y = np.zeros(2 ** int(st.sum()))
        _lib.check(self.lib.mmhn_kron_diag(self.h, ltp, sp, y.ctypes.data_as(f64p)))
        return y

    def diag_scal(self, log_d, state, p, which):
        d, dp_ = _f(log_d); pv, pp = _f(p); st, sp = _s(state)
        y = np.zeros_like(pv)
        _lib.check(self.lib.mmhn_diag_scal(self.h, dp_, sp, pp, y.ctypes.data_as(f64p), int(which)))
        return y

    def obs_indices(self, state, pt_first):
        st, sp = _s(state)
        n = (st.shape[0] - 1) // 2
        free = int(st[1:2 * n:2].sum()) if pt_first else int(st[0:2 * n:2].sum())
        idx = np.zeros(2 ** free, dtype=np.int64)
        cnt = C.c_int64()
        _lib.check(self.lib.mmhn_obs_states(self.h, sp, int(bool(pt_first)), idx.ctypes.data_as(i64p), C.byref(cnt)))
        return idx[:cnt.value]

    def resolvent(self, log_theta, log_d_p, log_d_m, x, state, transpose=False):
        keep, (a, b, c) = self._params(log_theta, log_d_p, log_d_m)
        xv, xp = _f(x); st, sp = _s(state)
        y = np.zeros_like(xv)
        _lib.check(self.lib.mmhn_resolvent(self.h, a, b, c, sp, xp, y.ctypes.data_as(f64p), int(transpose)))
        return y

    def x_partial_Q_y(self, log_theta, x, y, state):
        lt, ltp = _f(log_theta); xv, xp = _f(x); yv, yp = _f(y); st, sp = _s(state)
        G = np.zeros((self.N, self.N))
        _lib.check(self.lib.mmhn_x_partial_Q_y(self.h, ltp, sp, xp, yp, G.ctypes.data_as(f64p)))
        return G

    def x_partial_D_y(self, log_d_p, log_d_m, state, x, y):
        a, ap = _f(log_d_p); b, bp = _f(log_d_m); xv, xp = _f(x); yv, yp = _f(y); st, sp = _s(state)
        ddp, ddm = np.zeros(self.N), np.zeros(self.N)
        _lib.check(self.lib.mmhn_x_partial_D_y(self.h, ap, bp, sp, xp, yp, ddp.ctypes.data_as(f64p),
                                               ddm.ctypes.data_as(f64p)))
        return ddp, ddm

    def partial_diag_scal(self, log_d, state, p, i, which):
        d, dp_ = _f(log_d); pv, pp = _f(p); st, sp = _s(state)
        y = np.zeros_like(pv)
        _lib.check(self.lib.mmhn_partial_diag_scal(self.h, dp_, sp, pp, int(i), int(which), y.ctypes.data_as(f64p)))
        return y

    # ---- single-tumour primitives
    def v_kronvec(self, log_theta, p, state, diag=True, transpose=False):
        lt, ltp = _f(log_theta); pv, pp = _f(p); st, sp = _s(state)
        y = np.zeros_like(pv)
        _lib.check(self.lib.mmhn_v_kronvec(self.h, ltp, sp, pp, y.ctypes.data_as(f64p), int(diag), int(transpose)))
        return y

    def v_resolvent(self, log_theta, x, state, d_rates=None, transpose=False):
        lt, ltp = _f(log_theta); xv, xp = _f(x); st, sp = _s(state)
        y = np.zeros_like(xv)
        if d_rates is None or np.isscalar(d_rates):
            if d_rates is not None and float(d_rates) != 1.0:
                d_rates = np.full_like(xv, float(d_rates))
            else:
                d_rates = None
        dr = None
        if d_rates is not None:
            dkeep, dr = _f(d_rates)
        _lib.check(self.lib.mmhn_v_resolvent(self.h, ltp, sp, dr, xp, y.ctypes.data_as(f64p), int(transpose)))
        return y

    def v_x_partial_Q_y(self, log_theta, x, y, state):
        lt, ltp = _f(log_theta); xv, xp = _f(x); yv, yp = _f(y); st, sp = _s(state)
        G, dd = np.zeros((self.N, self.N)), np.zeros(self.N)
        _lib.check(self.lib.mmhn_v_x_partial_Q_y(self.h, ltp, sp, xp, yp, G.ctypes.data_as(f64p),
                                                 dd.ctypes.data_as(f64p)))
        return G, dd

    def v_kron_diag(self, log_theta, state, diag=None):
        lt, ltp = _f(log_theta); st, sp = _s(state)
        out = np.zeros(2 ** int(st.sum()))
        dg = None
        if diag is not None:
            dkeep, dg = _f(diag)
        _lib.check(self.lib.mmhn_v_kron_diag(self.h, ltp, sp, dg, out.ctypes.data_as(f64p)))
        return out

    def v_scal_d_pt(self, log_d_p, log_d_m, state, vec):
        a, ap = _f(log_d_p); b, bp = _f(log_d_m); v, vp = _f(vec); st, sp = _s(state)
        op, om = np.zeros_like(v), np.zeros_like(v)
        _lib.check(self.lib.mmhn_v_scal_d_pt(self.h, ap, bp, sp, vp, op.ctypes.data_as(f64p), om.ctypes.data_as(f64p)))
        return op, om

    def v_d_scal_d_pt(self, log_d_p, log_d_m, state, vec, i):
        a, ap = _f(log_d_p); b, bp = _f(log_d_m); v, vp = _f(vec); st, sp = _s(state)
        op, om = np.zeros_like(v), np.zeros_like(v)
        _lib.check(self.lib.mmhn_v_d_scal_d_pt(self.h, ap, bp, sp, vp, int(i), op.ctypes.data_as(f64p),
                                               om.ctypes.data_as(f64p)))
        return op, om

    def v_x_partial_D_y(self, log_d_p, log_d_m, state, x, y):
        a, ap = _f(log_d_p); b, bp = _f(log_d_m); xv, xp = _f(x); yv, yp = _f(y); st, sp = _s(state)
        ddp, ddm = np.zeros(self.N), np.zeros(self.N)
        _lib.check(self.lib.mmhn_v_x_partial_D_y(self.h, ap, bp, sp, xp, yp, ddp.ctypes.data_as(f64p),
                                                 ddm.ctypes.data_as(f64p)))
        return ddp, ddm

    # ---- patient shards on several GPUs
    def comm_init(self, unique_id: bytes, rank: int, n_ranks: int):
        """Join the RCCL communicator `unique_id` (from `unique_id()` on rank 0); afterwards cohort_sums / score /
        score_and_grad return the sums over all ranks (one all-reduce on the engine's stream per call)."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _lib.check(self.lib.mmhn_comm_init(self.h, buf, int(rank), int(n_ranks)))
        self.comm_size = int(n_ranks)

    def comm_destroy(self):
        _lib.check(self.lib.mmhn_comm_destroy(self.h))
        self.comm_size = 1

    # ---- simulation
    def simulate(self, log_theta, pt_d_ef, mt_d_ef, n_sim, seed=0, orders=False):
        """Gillespie samples: int8 dat [n_sim, 2n+2] (and the event sequences [n_sim, 2N+2] if `orders`)."""
        lt, ltp = _f(log_theta); a, ap = _f(pt_d_ef); b, bp = _f(mt_d_ef)
        n_sim = int(n_sim)
        dat = np.zeros((n_sim, 2 * self.n + 2), dtype=np.int8)
        od = np.zeros((n_sim, 2 * self.N + 2), dtype=np.int8) if orders else None
        _lib.check(self.lib.mmhn_simulate(self.h, ltp, ap, bp, n_sim, int(seed) & (2 ** 64 - 1),
                                          dat.ctypes.data_as(_lib.i8p),
                                          od.ctypes.data_as(_lib.i8p) if orders else None))
        return (dat, od) if orders else dat

    # ---- measurement
    def bench_kronvec(self, log_theta, state, batch, iters, transpose=False, jacobi=False, tiles=False):
        """ms per launch of mmhn_kronvec_batched's launch (or the fused Jacobi step); tiles=True also returns
        (tiles with entries of Q_off, tiles per launch)."""
        lt, ltp = _f(log_theta); st, sp = _s(state)
        ms = C.c_double()
        tl = (C.c_int64 * 2)()
        _lib.check(self.lib.mmhn_bench_kronvec(self.h, ltp, sp, int(batch), int(iters), int(transpose), int(jacobi),
                                               C.byref(ms), tl))
        return (ms.value, int(tl[0]), int(tl[1])) if tiles else ms.value

    def bench_stream(self, nbytes=1 << 30, iters=10, kind="copy"):
        """Measured device-memory bandwidth (GB/s) of a plain copy / triad stream on this GPU."""
        out = C.c_double()
        _lib.check(self.lib.mmhn_bench_stream(self.h, int(nbytes), int(iters), 0 if kind == "copy" else 1, C.byref(out)))
        return out.value

    def counters(self):
        c = _lib.Counters()
        _lib.check(self.lib.mmhn_get_counters(self.h, C.byref(c)))
        out = {"eval_ms": c.eval_ms, "evals": c.evals, "comm_ranks": int(c.comm_ranks), "comm_rank": int(c.comm_rank)}
        for i, name in enumerate(_lib.KERNEL_CLASSES):
            k = c.kernel[i]
            out[name] = {"ms": k.ms, "launches": k.launches, "alg_bytes": k.alg_bytes}
        return out

    def reset_counters(self):
        _lib.check(self.lib.mmhn_reset_counters(self.h))

    def debug_lane_moves(self, transposed=False):
        """out[i][lane] = source lane of the window solve's exchange along lane bit i (csrc/wsolve.h)."""
        out = (C.c_int * (6 * 64))()
        _lib.check(self.lib.mmhn_debug_lane_moves(self.h, int(bool(transposed)), out))
        return np.array(out[:], dtype=np.int64).reshape(6, 64)
