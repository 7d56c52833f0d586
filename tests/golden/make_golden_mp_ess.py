#!/usr/bin/env python3
"""Golden vectors for the oracle's mixed-precision and chain-statistics
restatements (oracle/mp_path.c, oracle/ess.c), the checkers of the GPU tests
in tests/test_gpu_mp*.py and tests/test_gpu_stats.py.  Like oracle_golden.json
they pin the ORACLE as validated, not a reference binary (none can be built
here).  Floats are C99 hex strings.

    python tests/golden/make_golden_mp_ess.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_py as O  # noqa: E402
from mcmc_amd.likelihood import synthetic_host  # noqa: E402


def hx(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def mp_sequence(L, N, d, block_rows, rf, calls, seed, eps):
    """The state-carrying call sequence of tests/test_gpu_mp.py on a small data set."""
    X = synthetic_host(99, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    blocks = N // block_rows
    z = np.ones(blocks, dtype=np.int8)
    rng = np.random.default_rng(0)
    vals, bright, thetas = [], [], []
    for call in range(calls):
        th = rng.normal(0, 0.5, d)
        nb, nd = C.c_int(), C.c_int()
        v = L.oracle_mp_likelihood(O.ptr(X), O.ptr(Xf), N, d, N, O.ptr(th), block_rows, blocks, rf, call % rf, seed,
                                   call, eps, O.ptr(z), C.byref(nb), C.byref(nd))
        vals.append(v)
        bright.append(nb.value)
        thetas.append(hx(th))
    return {"N": N, "d": d, "data_seed": 99, "block_rows": block_rows, "blocks": blocks, "resamp_factor": rf,
            "u_seed": seed, "eps": float(eps).hex(), "theta": thetas, "value": hx(vals), "bright_after": bright,
            "z_final_sum": int(z.sum())}


def ar1(n, d, rho, seed):
    rng = np.random.default_rng(seed)
    x = np.zeros((n, d))
    e = rng.normal(size=(n, d))
    for s in range(1, n):
        x[s] = rho * x[s - 1] + e[s]
    return x


def main():
    L = O.lib()
    L.oracle_set_verbose(0)
    L.oracle_mp_likelihood.restype = C.c_double
    L.oracle_mp_likelihood.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.oracle_mp_bound.restype = C.c_double
    L.oracle_mp_bound.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.POINTER(C.c_double)]
    out = {"mp_sequences": [mp_sequence(L, 4096, 8, 64, 2, 6, 1234, 1.7e-7),
                            mp_sequence(L, 5000, 4, 128, 3, 5, 77, 1.7e-7)]}

    X = synthetic_host(5, 3000, 6)
    Xf = np.asfortranarray(X.astype(np.float32))
    th = np.random.default_rng(1).normal(0, 1.0, 6)
    eps = C.c_double()
    s = L.oracle_mp_bound(O.ptr(X), O.ptr(Xf), 3000, 6, 3000, O.ptr(th), C.byref(eps))
    out["mp_bound"] = {"N": 3000, "d": 6, "data_seed": 5, "theta": hx(th), "sum_sigmoid": float(s).hex(),
                       "epsilon": float(eps.value).hex()}

    x = ar1(700, 5, 0.9, 3)
    ac = np.zeros(60)
    tot = L.circular_autocorrelation(O.ptr(ac), O.ptr(x), 5, 700, 60)
    out["autocorrelation"] = {"n": 700, "d": 5, "rho": 0.9, "seed": 3, "lags": 60, "out_v": hx(ac),
                              "sum": float(tot).hex(), "ess": float(700 / (1 + 2 * tot)).hex()}

    N, d = 400, 3
    X = synthetic_host(12345, N, d)
    mct = O.mcmc_tune_str(rwsd=0.5, rwsdf=0.5)
    rng = O.Rng(7, L)
    mvout = np.empty(N)
    init = np.zeros(d)
    L.tune_ess_cpu(O.data_str(data=O.ptr(X).value, mvout=O.ptr(mvout).value), rng.r,
                   O.mcmc_str(ddata=d, Nd=N, Ns=1, burnin=50, impl=0), C.byref(mct), O.ptr(init), 600)
    out["tune_ess"] = {"N": N, "d": d, "data_seed": 12345, "chain_seed": 7, "burnin": 50, "length": 600,
                       "rwsd_in": 0.5, "rwsd_out": float(mct.rwsd).hex()}

    path = os.path.join(HERE, "oracle_golden_mp_ess.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()
