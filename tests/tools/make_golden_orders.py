#!/usr/bin/env python3
"""Golden vectors for order likelihoods and likeliest orders (metmhn/model.py).

Runs ONLY in the build container (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tests/tools/jax_standin:/root/reference \
    python tests/tools/make_golden_orders.py

The reference's `metmhn.model` needs two things this image lacks besides jax:
  * its Cython helper `int_order_conversion` -- compiled from the reference's own .pyx, where it
    lies, into a scratch directory under $TMPDIR (nothing of it enters this repository);
  * the PyPI package `mhn` (imported for `oMHN`, used only by met_status "absent"/"present") --
    absent, so an import-only placeholder is put on the path whose every attribute raises.  NO
    value written here can depend on it: the "absent"/"present" cases are not generated (the tests
    anchor them on closed-form expansions of the observation-MHN path probability instead).

Writes tests/golden/orders.npz (data only: inputs + the reference's outputs).  Per model m (n = 4, 5):
  m{m}_theta, m{m}_obs1, m{m}_obs2                the parameters
  m{m}_du_state [C,n+1], m{m}_du_seed [C], m{m}_du_off [C+1], m{m}_du_val   _get_diag_unpaired
  m{m}_lk_kind [C] (0 isMetastasis, 1 PT, 2 Met, 3 unknown, 4 sync), m{m}_lk_order [C,2n+1] (-1 padded),
  m{m}_lk_p [C]                                    likelihood(order, ...)
  m{m}_lo_kind [C], m{m}_lo_state [C,2n+1], m{m}_lo_order [C,2n+1], m{m}_lo_p [C], m{m}_lo_sec [C]
                                                   likeliest_order(state, ...) and its run time
"""
import os
import subprocess
import sys
import tempfile
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "golden", "orders.npz")
REF = "/root/reference"
KINDS = ["isMetastasis", "PT", "Met", "unknown", "sync"]


def scratch_imports():
    tmp = tempfile.mkdtemp(prefix="mmhn_orders_")
    ext = os.path.join(tmp, "ioc")
    os.makedirs(ext)
    with open(os.path.join(ext, "setup.py"), "w") as f:
        f.write("from setuptools import setup, Extension\nfrom Cython.Build import cythonize\n"
                f"setup(ext_modules=cythonize([Extension('int_order_conversion', ['{REF}/metmhn/"
                f"int_order_conversion.pyx'])], language_level=3, build_dir='{ext}/gen'))\n")
    subprocess.run([sys.executable, "setup.py", "-q", "build_ext", "--build-lib", ext, "--build-temp",
                    ext + "/obj"], cwd=ext, check=True, stdout=subprocess.DEVNULL)
    stub = os.path.join(tmp, "absent", "mhn")
    os.makedirs(stub)
    open(os.path.join(stub, "__init__.py"), "w").close()
    with open(os.path.join(stub, "model.py"), "w") as f:
        f.write("class oMHN:\n    def __init__(self, *a, **k):\n        pass\n"
                "    def __getattr__(self, name):\n"
                "        raise RuntimeError('PyPI mhn is absent from this container')\n")
    sys.path.append(os.path.dirname(stub))
    import metmhn
    metmhn.__path__.append(ext)


def random_model(rng, n):
    th = rng.normal(0.0, 0.6, (n + 1, n + 1))
    th[np.diag_indices(n + 1)] = rng.normal(-1.5, 0.5, n + 1)
    obs1 = rng.normal(0.0, 0.4, n + 1)
    obs2 = rng.normal(0.0, 0.4, n + 1)
    return th, obs1, obs2


def random_paired_state(rng, n, k_max):
    while True:
        pt = rng.random(n) < 0.55
        mt = rng.random(n) < 0.55
        st = np.zeros(2 * n + 1, dtype=bool)
        st[0:2 * n:2], st[1:2 * n:2], st[2 * n] = pt, mt, True
        if 3 <= st.sum() <= k_max:
            return st


def random_order(rng, st, n):
    """A random order the chain can take to `st` (seeded): joint events first, the seeding, the rest."""
    pt, mt = st[0:2 * n:2], st[1:2 * n:2]
    both = [i for i in range(n) if pt[i] and mt[i]]
    pre = [i for i in both if rng.random() < 0.5]
    rng.shuffle(pre)
    order = []
    for i in pre:
        order += [2 * i, 2 * i + 1]
    order.append(2 * n)
    rest = [2 * i for i in range(n) if pt[i] and i not in pre] + [2 * i + 1 for i in range(n) if mt[i] and i not in pre]
    rng.shuffle(rest)
    return order + rest


def pad(order, n):
    o = -np.ones(2 * n + 1, dtype=np.int64)
    o[:len(order)] = order
    return o


def main():
    warnings.filterwarnings("ignore")
    scratch_imports()
    from metmhn.model import MetMHN
    from metmhn.state import MetState, State

    rng = np.random.default_rng(20240607)
    out = {}
    for m, (n, k_max, n_states) in enumerate([(4, 9, 10), (5, 10, 8)]):
        th, obs1, obs2 = random_model(rng, n)
        mod = MetMHN(th, obs1, obs2)
        pre = f"m{m}_"
        out[pre + "theta"], out[pre + "obs1"], out[pre + "obs2"] = th, obs1, obs2

        du_state, du_seed, du_val, du_off = [], [], [], [0]
        for _ in range(6):
            seed = bool(rng.random() < 0.5)
            size = n + 1 if seed else n
            s = rng.random(size) < 0.6
            d = mod._get_diag_unpaired(State.from_seq(s), seeding=seed)
            du_state.append(np.concatenate((s, np.zeros(n + 1 - size, dtype=bool))))
            du_seed.append(seed)
            du_val.append(np.asarray(d))
            du_off.append(du_off[-1] + d.size)
        out[pre + "du_state"], out[pre + "du_seed"] = np.array(du_state), np.array(du_seed)
        out[pre + "du_val"], out[pre + "du_off"] = np.concatenate(du_val), np.array(du_off)

        lk_kind, lk_order, lk_p = [], [], []
        lo_kind, lo_state, lo_order, lo_p, lo_sec = [], [], [], [], []

        # unpaired metastases
        for _ in range(5):
            while True:
                mt = rng.random(n) < 0.6
                if mt.sum() >= 2:
                    break
            st = np.zeros(2 * n + 1, dtype=bool)
            st[1:2 * n:2], st[2 * n] = mt, True
            t0 = time.perf_counter()
            o, p = mod.likeliest_order(MetState.from_seq(st), "isMetastasis")
            lo_sec.append(time.perf_counter() - t0)
            lo_kind.append(0), lo_state.append(st), lo_order.append(pad(o, n)), lo_p.append(p)
            ev = [2 * i + 1 for i in range(n) if mt[i]] + [2 * n]
            for _ in range(2):
                rng.shuffle(ev)
                lk_kind.append(0), lk_order.append(pad(ev, n))
                lk_p.append(mod.likelihood(tuple(ev), "isMetastasis"))

        # paired samples
        for _ in range(n_states):
            st = random_paired_state(rng, n, k_max)
            for kind in (1, 2, 3, 4):
                t0 = time.perf_counter()
                o, p = mod.likeliest_order(MetState.from_seq(st), "isPaired", KINDS[kind])
                lo_sec.append(time.perf_counter() - t0)
                lo_kind.append(kind), lo_state.append(st), lo_order.append(pad(o, n)), lo_p.append(p)
                again = mod.likelihood(tuple(int(e) for e in o), "isPaired", KINDS[kind])
                assert abs(again - p) <= 1e-12 * abs(p), (kind, o, p, again)
                for _ in range(2):
                    ro_ = random_order(rng, st, n)
                    lk_kind.append(kind), lk_order.append(pad(ro_, n))
                    lk_p.append(mod.likelihood(tuple(ro_), "isPaired", KINDS[kind]))
            print(f"n={n} state {st.astype(int)} done", flush=True)

        out[pre + "lk_kind"], out[pre + "lk_order"], out[pre + "lk_p"] = \
            np.array(lk_kind), np.array(lk_order), np.array(lk_p, dtype=np.float64)
        out[pre + "lo_kind"], out[pre + "lo_state"], out[pre + "lo_order"] = \
            np.array(lo_kind), np.array(lo_state), np.array(lo_order)
        out[pre + "lo_p"], out[pre + "lo_sec"] = np.array(lo_p, dtype=np.float64), np.array(lo_sec)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
