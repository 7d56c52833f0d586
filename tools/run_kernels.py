#!/usr/bin/env python3
"""Launch every likelihood kernel a few times at BASELINE sizes (for rocprofv3
--pmc passes): SP and DP streaming, batched f32 / f64, mixed precision single
chain and K chains."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import Likelihood  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(0)
N = 10_000_000
with Likelihood(N, 64, precision=3, segments=1) as lk:  # DP, SP, MP single at d=64
    lk.fill_synthetic(12345)
    th = rng.normal(0, 0.01, 64)
    for _ in range(reps):
        lk.loglik(th)
        lk.loglik_f32(th.astype(np.float32))
    blocks = N // 64
    lk.mp_reset(64, blocks)
    for call in range(reps + 1):
        lk.mp_loglik(th, 64, blocks, 16, call % 16, 1234, call, 1.7e-7)
with Likelihood(N, 128, precision=3, segments=1) as lk:  # batched at d=128, K=64
    lk.fill_synthetic(12345)
    Th = rng.normal(0, 0.01, (64, 128))
    for _ in range(reps):
        lk.set_batched_f32_mode("exact")  # f32 MFMA
        lk.loglik_batched(Th.astype(np.float32), dtype=np.float32)
        lk.set_batched_f32_mode("split")  # three bf16 pieces
        lk.loglik_batched(Th.astype(np.float32), dtype=np.float32)
        lk.set_batched_f64_mode("fast")  # product-form epilogue (default)
        lk.loglik_batched(Th, dtype=np.float64)
    lk.set_batched_f64_mode("exact")  # per-element series: the other instantiation of lik_batched_f64_kernel
    for _ in range(2):
        lk.loglik_batched(Th, dtype=np.float64)
    lk.set_batched_f64_mode("fast")
    blocks = N // 64
    seeds = np.full(64, 1234, dtype=np.uint64)
    for mode in ("exact", "split"):
        lk.set_batched_f32_mode(mode)
        lk.mp_reset_batched(64, blocks, 64)
        for call in range(reps + 1):
            lk.mp_loglik_batched(Th, 16, np.full(64, call % 16), seeds, call, 1.7e-7)
    # one phase per chain (K independent mp_sampler runs): the per-(block, chain) correction kernel, mpb_fix_kernel
    lk.mp_reset_batched(64, blocks, 64)
    for call in range(reps + 1):
        lk.mp_loglik_batched(Th, 16, (np.arange(64) + call) % 16, seeds, call, 1.7e-7)
print("done")
