#!/usr/bin/env python3
"""Time the batched (MFMA) likelihood: python tools/bench_batched.py [N] [d] [K] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import Likelihood  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
for prec, dt, name in ((2, np.float32, "f32"), (1, np.float64, "f64")):
    with Likelihood(N, d, precision=prec, timing=True) as lk:
        lk.fill_synthetic(12345)
        Th = np.random.default_rng(0).normal(0, 0.01, (K, d)).astype(dt)
        ms = []
        for r in range(reps + 2):
            t0 = time.perf_counter()
            out = lk.loglik_batched(Th, dtype=dt)
            wall = (time.perf_counter() - t0) * 1e3
            k_ms, nbytes = lk.last_kernel()
            if r >= 2:
                ms.append((k_ms, wall))
        k_ms = float(np.median([m[0] for m in ms]))
        wall = float(np.median([m[1] for m in ms]))
        flops = 2.0 * N * d * K
        print("%s N=%d d=%d K=%d: kernel %.3f ms (wall %.3f)  %.1f TFLOP/s  %.1f GB/s  chain-iterations/s %.0f  L0=%.6g" %
              (name, N, d, K, k_ms, wall, flops / k_ms / 1e9, nbytes / k_ms / 1e6, K / (wall * 1e-3), out[0]))
