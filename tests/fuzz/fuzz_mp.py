#!/usr/bin/env python3
"""Random shapes of the mixed-precision K-chain evaluation (dark pass in both float modes; dense and per-(block, chain)
correction) against the single-chain HIP path chain by chain: identical indicator counts, values within 2e-6 relative
plus one float rounding (1e-7) per row -- a float sum of terms near zero is only that good in either path
(checker run by hand on the GPU box, not collected by pytest).  python tests/fuzz/fuzz_mp.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mcmc_amd import Likelihood  # noqa: E402


def one_case(rng, case):
    N = int(np.exp(rng.uniform(np.log(64), np.log(400000))))
    d = int(np.exp(rng.uniform(0, np.log(256))))
    if N * d > 20_000_000:
        N = max(64, 20_000_000 // d)
    block_rows = int(rng.choice([64, 128, 256, 512, 1024]))
    full = max(1, N // block_rows)
    blocks = int(rng.choice([full, max(1, full - int(rng.integers(0, 3))), full + int(rng.integers(0, 3))]))
    K = int(rng.integers(1, 65))
    rf = int(rng.integers(1, 9))
    shared = bool(rng.integers(0, 2))
    mode = str(rng.choice(["split", "exact"]))
    eps = float(rng.choice([1.7e-7, 1e-6]))
    scale = float(rng.choice([0.2, 1.0, 4.0])) / np.sqrt(d)
    tag = "case %d N=%d d=%d block_rows=%d blocks=%d K=%d rf=%d shared=%d %s eps=%g scale=%.2g" % (
        case, N, d, block_rows, blocks, K, rf, shared, mode, eps, scale * np.sqrt(d))
    calls = 3
    Ths = [rng.normal(0, scale, (K, d)) for _ in range(calls)]
    chooses = [rng.integers(0, rf, K).astype(np.int32) for _ in range(calls)]
    if shared:
        for c in chooses:
            c[:] = c[0]
    seeds = rng.integers(1, 1 << 40, K).astype(np.uint64)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.fill_synthetic(int(rng.integers(1, 1 << 30)))
        lk.set_batched_f32_mode(mode)
        lk.mp_reset_batched(block_rows, blocks, K)
        batched = [lk.mp_loglik_batched(Ths[c], rf, chooses[c], seeds, c, eps) for c in range(calls)]
        for k in sorted(set(int(v) for v in rng.integers(0, K, 3))):
            lk.mp_reset(block_rows, blocks)
            for c in range(calls):
                v, nb, nd = lk.mp_loglik(Ths[c][k], block_rows, blocks, rf, int(chooses[c][k]), int(seeds[k]), c, eps)
                bv, bb, bd = batched[c]
                assert (bb[k], bd[k]) == (nb, nd), (tag, k, c, bb[k], bd[k], nb, nd)
                ok = (np.isnan(v) and np.isnan(bv[k])) or (np.isinf(v) and v == bv[k]) or abs(bv[k] - v) <= 2e-6 * abs(v) + 1e-7 * min(N, blocks * block_rows)
                assert ok, (tag, k, c, bv[k], v)
    return tag


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for c in range(cases):
        print(one_case(rng, c), flush=True)
    print("fuzz_mp: %d cases clean" % cases)


if __name__ == "__main__":
    main()
