#!/usr/bin/env python3
"""Random shapes through the C-ABI against the CPU restatement (checker run by hand on the GPU box, not collected by pytest; uses tests/oracle_py.py as
the checker): single evaluations in double and float, K-chain passes in all float / double modes, and the
device-resident chain in its one-launch and two-kernel forms against each other and against a host loop over single
evaluations.  python tests/fuzz/fuzz_shapes.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from mcmc_amd import Likelihood  # noqa: E402
from mcmc_amd.likelihood import synthetic_host  # noqa: E402


def one_case(rng, case):
    N = int(np.exp(rng.uniform(0, np.log(3000000))))
    d = int(np.exp(rng.uniform(0, np.log(300))))
    if N * d > 40_000_000:
        N = max(1, 40_000_000 // d)
    segments = int(rng.choice([1, 2, 8, 16]))
    K = int(rng.integers(1, 65))
    X = synthetic_host(int(rng.integers(1, 1 << 30)), N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    scale = rng.choice([0.05, 0.5, 3.0]) / np.sqrt(d)
    theta = rng.normal(0, scale, d)
    tag = "case %d N=%d d=%d segments=%d K=%d scale=%.3g" % (case, N, d, segments, K, scale * np.sqrt(d))
    want, _ = O.log_likelihood(X, theta)
    with Likelihood(N, d, precision=3, segments=segments) as lk:
        lk.upload(X)
        lk.upload(Xf)
        got = lk.loglik(theta)
        assert abs(got - want) <= 1e-10 * abs(want) + 1e-12, (tag, got, want)
        gotf = lk.loglik(theta.astype(np.float32))
        assert abs(gotf - want) <= 1e-4 * abs(want) + 1e-4, (tag, gotf, want)
        # K chains
        Th = rng.normal(0, scale, (K, d))
        wantK = np.array([O.log_likelihood(X, Th[k])[0] for k in range(K)])
        for mode in ("fast", "exact"):
            lk.set_batched_f64_mode(mode)
            gk = lk.loglik_batched(Th)
            assert np.all(np.abs(gk - wantK) <= 1e-10 * np.abs(wantK) + 1e-12), (tag, mode, np.max(np.abs(gk - wantK) / np.abs(wantK)))
        for mode in ("split", "exact"):
            lk.set_batched_f32_mode(mode)
            gk = lk.loglik_batched(Th.astype(np.float32), dtype=np.float32)
            assert np.all(np.abs(gk - wantK) <= 2e-4 * np.abs(wantK) + 1e-4), (tag, mode, np.max(np.abs(gk - wantK) / np.abs(wantK)))
        # device-resident chain: one launch / two kernels / host loop over single evaluations
        n = int(rng.integers(1, 40))
        inc = rng.normal(0, 0.3 * scale, (n, d))
        logu = np.log(rng.uniform(size=n))
        runs = {}
        for persist in ("1", "0"):
            os.environ["MHLIK_CHAIN_PERSIST"] = persist
            cp = lk.chain_begin(theta, 10.0)
            r = lk.chain_run(inc, logu)
            runs[persist] = (cp, r, lk.last_kernel_name())
        os.environ.pop("MHLIK_CHAIN_PERSIST")
        a, b = runs["1"], runs["0"]
        assert a[0] == b[0] and np.array_equal(a[1]["lik"], b[1]["lik"]) and np.array_equal(a[1]["samples"], b[1]["samples"]) and \
            np.array_equal(a[1]["accepted"], b[1]["accepted"]) and np.array_equal(a[1]["ratio"], b[1]["ratio"]), (tag, a[2], b[2])
        cur = theta.copy()
        for it in range(n):
            prop = cur + inc[it]
            Lp = lk.loglik(prop)
            assert Lp == b[1]["lik"][it], (tag, it, Lp, b[1]["lik"][it])
            if b[1]["accepted"][it]:
                cur = prop
            assert np.array_equal(cur, b[1]["samples"][it]), (tag, it)
    return tag + "  " + a[2].split("<")[0]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for c in range(cases):
        print(one_case(rng, c), flush=True)
    print("fuzz: %d cases clean" % cases)


if __name__ == "__main__":
    main()
