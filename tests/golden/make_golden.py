#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the CPU oracle.

The reference commits no outputs and cannot be built here (GSL, CUDA absent),
so these vectors pin the ORACLE (oracle/liboracle.so), not a reference binary:
they freeze what the oracle produced when it was validated, so that a later
change to the oracle, the RNG shim or the data generator shows up as a diff.
Floats are stored as C99 hex strings (exact).

    python tests/golden/make_golden.py
"""
import hashlib
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


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    L = O.lib()
    out = {}

    # (i) MT19937 known answers: public vector for seed 5489 and GSL's own self-test value
    r = O.Rng(5489, L)
    first = [L.gsl_rng_get(r.r) for _ in range(10000)]
    r.set(4357)
    gsl = [L.gsl_rng_get(r.r) for _ in range(1000)]
    r.set(0)
    seed0 = [L.gsl_rng_get(r.r) for _ in range(5)]
    out["mt19937"] = {"seed5489_first5": first[:5], "seed5489_10000th": first[-1], "seed4357_1000th": gsl[-1],
                      "seed0_first5": seed0,
                      "source": "Matsumoto & Nishimura mt19937ar.c output for seed 5489 (first value 3499211612, "
                                "10000th 4123659995); GSL rng/test.c: rng_test(gsl_rng_mt19937, 4357, 1000, 1186927261)"}
    # GSL's reference manual ("Random number environment variables" / the uniform
    # example) prints, for the default generator mt19937 and seed 0: first value
    # 4293858116 and the uniforms 0.99974 0.16291 0.28262 0.94720 0.23166 0.48497
    # 0.95748 0.74431 0.54004 0.73995.
    out["gsl_manual_seed0"] = {"first_value": 4293858116,
                               "uniform_5dp": ["0.99974", "0.16291", "0.28262", "0.94720", "0.23166", "0.48497",
                                               "0.95748", "0.74431", "0.54004", "0.73995"]}
    r.set(42)
    out["ziggurat_seed42_first16"] = hx([L.gsl_ran_gaussian_ziggurat(r.r, 1.0) for _ in range(16)])
    r.set(42)
    out["uniform_seed42_first8"] = hx([L.gsl_rng_uniform(r.r) for _ in range(8)])

    # (ii) seeded chains on the synthetic data set
    chains = []
    for d in (3, 16):
        X = synthetic_host(12345, 10000, d)
        for seed in (0, 1, 2):
            rwsd = 2.38 / np.sqrt(d) * 0.05
            res = O.cpu_sampler(X, seed, 44, 21, rwsd)
            tr = res["trace"]
            chains.append({"N": 10000, "d": d, "data_seed": 12345, "chain_seed": seed, "Ns": 44, "burnin": 21,
                           "rwsd": float(rwsd).hex(), "x_sha256": sha(X),
                           "lik": hx(tr["lik"]), "ratio": hx(tr["ratio"]), "logu": hx(tr["logu"]),
                           "accept": [int(v) for v in tr["accept"]],
                           "last_sample": hx(res["samples"][-1]), "acceptance": float(res["acceptance"]).hex()})
    out["chains"] = chains

    # (iii) per-row dots for one theta at N=1024, d=64
    X = synthetic_host(777, 1024, 64)
    th = np.random.default_rng(5).normal(0, 0.2, 64)
    val, mvout = O.log_likelihood(X, th)
    out["rowdots"] = {"N": 1024, "d": 64, "data_seed": 777, "theta": hx(th), "x_sha256": sha(X), "mvout": hx(mvout),
                      "loglik": float(val).hex()}

    # (iv) edge cases
    edges = []
    for name, N, d, thv in [("d1", 513, 1, [0.7]), ("d128", 300, 128, None), ("ragged", 1000003 // 100, 5, None),
                            ("single_row", 1, 4, None)]:
        X = synthetic_host(31, N, d)
        th = np.array(thv) if thv is not None else np.random.default_rng(N).normal(0, 0.3, d)
        val, _ = O.log_likelihood(X, th)
        edges.append({"name": name, "N": N, "d": d, "data_seed": 31, "theta": hx(th), "loglik": float(val).hex()})
    Xo = np.zeros((1000, 2), order="F")
    Xo[:, 0] = 1.0
    Xo[:, 1] = np.linspace(-1, 1, 1000)
    for thv in ([800.0, 0.0], [-800.0, 0.0], [700.0, 5.0], [0.0, 0.0]):
        val, _ = O.log_likelihood(Xo, np.array(thv))
        edges.append({"name": "overflow", "N": 1000, "d": 2, "theta": hx(thv), "loglik": float(val).hex()})
    out["edges"] = edges

    with open(os.path.join(HERE, "oracle_golden.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", os.path.join(HERE, "oracle_golden.json"))


if __name__ == "__main__":
    main()
