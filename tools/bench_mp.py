#!/usr/bin/env python3
"""Time the mixed-precision likelihood: single chain (mhlik_mp_loglik) and K
lock-step chains (mhlik_mp_loglik_batched).
  python tools/bench_mp.py single [N] [d] [reps]
  python tools/bench_mp.py batched [N] [d] [K] [reps]
Steady state: the first call (every block bright) is timed separately."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import Likelihood  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "batched"
EPS = 1.7e-7
BLOCK_ROWS = 64

if mode == "single":
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    blocks = N // BLOCK_ROWS
    with Likelihood(N, d, precision=3, segments=1, timing=True) as lk:
        lk.fill_synthetic(12345)
        rng = np.random.default_rng(0)
        for rf in (2, 16, 64):
            lk.mp_reset(BLOCK_ROWS, blocks)
            ms = []
            for call in range(reps + 2):
                th = rng.normal(0, 0.01, d)
                v, nb, nd = lk.mp_loglik(th, BLOCK_ROWS, blocks, rf, call % rf, 1234, call, EPS)
                k_ms, _ = lk.last_kernel()
                ms.append(k_ms)
            first, steady = ms[0], float(np.median(ms[2:]))
            alg = N * d * 4 + (N // rf) * d * 8  # float pass + double rows of the re-examined blocks
            print("single N=%d d=%d F=%d: first call (all bright) %.3f ms, steady %.3f ms, %.1f GB/s of %.2f GB "
                  "algorithmic (f32 all rows + f64 1/F of the rows), bright after = %d of %d, L=%.6g" %
                  (N, d, rf, first, steady, alg / steady / 1e6, alg / 1e9, nb, blocks, v), flush=True)
else:
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    blocks = N // BLOCK_ROWS
    with Likelihood(N, d, precision=3, segments=1, timing=True) as lk:
        lk.fill_synthetic(12345)
        rng = np.random.default_rng(0)
        seeds = np.arange(1, K + 1, dtype=np.uint64)
        for rf, shared in ((2, False), (16, False), (64, False), (2, True), (16, True), (64, True)):
            lk.mp_reset_batched(BLOCK_ROWS, blocks, K)
            ms, wall = [], []
            for call in range(reps + 2):
                Th = rng.normal(0, 0.01, (K, d))
                # the re-examination phase: one draw per chain (what K independent mp_sampler runs do)
                # or one draw shared by all chains of the lock-step iteration
                choose = np.full(K, rng.integers(0, rf)) if shared else rng.integers(0, rf, K)
                t0 = time.perf_counter()
                v, nb, nd = lk.mp_loglik_batched(Th, rf, choose, seeds, call, EPS)
                wall.append((time.perf_counter() - t0) * 1e3)
                k_ms, _ = lk.last_kernel()
                ms.append(k_ms)
            first, steady, w = ms[0], float(np.median(ms[2:])), float(np.median(wall[2:]))
            flops = 2.0 * N * d * K
            print("batched N=%d d=%d K=%d F=%d %s phase: first call (all bright) %.3f ms, steady %.3f ms (wall %.3f), "
                  "float pass alone = %.1f TFLOP/s equivalent, chain-iterations/s %.0f, mean bright after = %.1f of %d, "
                  "L0=%.6g" % (N, d, K, rf, "shared" if shared else "per-chain", first, steady, w, flops / steady / 1e9, K / (w * 1e-3), nb.mean(), blocks,
                                v[0]), flush=True)
        # reference points on the same data: plain float and double batched passes
        Th = rng.normal(0, 0.01, (K, d))
        for dt, name in ((np.float32, "f32"), (np.float64, "f64")):
            t = []
            for r in range(6):
                lk.loglik_batched(Th, dtype=dt)
                t.append(lk.last_kernel()[0])
            print("batched %s plain pass on the same context: %.3f ms" % (name, float(np.median(t[2:]))), flush=True)
