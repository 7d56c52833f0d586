#!/usr/bin/env python3
"""Time circular_autocorrelation: host C loop (chain_stats.c) vs device (mhlik_autocorrelation_f64).
   python tools/bench_stats.py [n] [d] [lags]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import Likelihood  # noqa: E402
from mcmc_amd import samplers as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lags = int(sys.argv[3]) if len(sys.argv) > 3 else 500
rng = np.random.default_rng(0)
x = np.cumsum(rng.normal(size=(n, d)), axis=0) * 0.01 + rng.normal(size=(n, d))
t0 = time.perf_counter()
ess_host = S.get_ess(x, lags)
t_host = time.perf_counter() - t0
with Likelihood(8, d) as lk:
    lk.autocorrelation(x, lags)  # warm-up (allocation)
    t0 = time.perf_counter()
    for _ in range(10):
        out, s = lk.autocorrelation(x, lags)
    t_dev = (time.perf_counter() - t0) / 10
print("autocorrelation n=%d d=%d lags=%d: host %.1f ms, device %.3f ms (incl. upload), ESS host %.6f device %.6f" %
      (n, d, lags, t_host * 1e3, t_dev * 1e3, ess_host, n / (1 + 2 * s)))
