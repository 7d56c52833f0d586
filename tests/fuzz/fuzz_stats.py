#!/usr/bin/env python3
"""Random shapes of the device autocorrelation against the CPU restatement of
circular_autocorrelation (oracle/ess.c): every lag within 1e-9 relative, the sum and the ESS it gives.  Run by hand on
the GPU box, not collected by pytest.  python tests/fuzz/fuzz_stats.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from mcmc_amd import Likelihood  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for c in range(cases):
        n = int(np.exp(rng.uniform(np.log(2), np.log(20000))))
        d = int(np.exp(rng.uniform(0, np.log(300))))
        if n * d > 2_000_000:
            n = max(2, 2_000_000 // d)
        lags = int(rng.integers(1, min(n, 2000) + 1))
        rho = float(rng.choice([0.0, 0.5, 0.9, 0.99]))
        x = np.zeros((n, d))
        e = rng.normal(size=(n, d))
        for s in range(1, n):
            x[s] = rho * x[s - 1] + e[s]
        x += rng.normal(size=d) * float(rng.choice([0.0, 3.0, 1e3]))
        want = np.zeros(lags)
        wsum = O.lib().circular_autocorrelation(O.ptr(want), O.ptr(x), d, n, lags)
        with Likelihood(8, d) as lk:
            got, gsum = lk.autocorrelation(x, lags)
        tag = "case %d n=%d d=%d lags=%d rho=%.2f" % (c, n, d, lags, rho)
        both_nan = np.isnan(want) & np.isnan(got)
        scale = np.maximum(np.abs(want), 1e-300)
        with np.errstate(invalid="ignore", over="ignore"):
            err = np.where(both_nan | (got == want), 0.0, np.abs(got - want) / scale)  # equal infinities count as equal
        assert np.all(err <= 1e-8) or np.all(np.abs(got - want)[~both_nan] <= 1e-12), (tag, float(np.nanmax(err)), int(np.isnan(err).sum()), got[np.isnan(err)][:3], want[np.isnan(err)][:3])
        if np.isfinite(wsum):
            assert abs(gsum - wsum) <= 1e-8 * max(1.0, abs(wsum)), (tag, gsum, wsum)
        print(tag, flush=True)
    print("fuzz_stats: %d cases clean" % cases)


if __name__ == "__main__":
    main()
