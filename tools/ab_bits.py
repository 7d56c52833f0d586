#!/usr/bin/env python3
"""Bitwise A/B of two builds of libmhlik.so on one box: the K-chain passes (float in both MFMA modes, double with both epilogues,
mixed precision with a shared phase and with per-chain phases; the exact float mode's dark pass too) on the same inputs; prints a digest per library (child process each), and whether they
agree.  usage: python tools/ab_bits.py libA.so libB.so"""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib):
    import mcmc_amd._lib as L
    L.lib_path = lambda name="libmhlik.so": lib
    from mcmc_amd import Likelihood
    h = hashlib.sha256()
    for N, d, K in ((1_000_003, 128, 64), (200_000, 37, 21)):
        rng = np.random.default_rng(N)
        Th = rng.normal(0, 0.02, (K, d))
        with Likelihood(N, d, precision=3, segments=1) as lk:
            lk.fill_synthetic(12345)
            for mode in ("split", "exact"):  # three bf16 pieces / f32 MFMA
                lk.set_batched_f32_mode(mode)
                h.update(lk.loglik_batched(Th.astype(np.float32), dtype=np.float32).tobytes())
            for mode in ("fast", "exact"):   # f64 MFMA, product-form / per-element epilogue
                lk.set_batched_f64_mode(mode)
                h.update(lk.loglik_batched(Th).tobytes())
            lk.set_batched_f32_mode("split")
            blocks = (N + 63) // 64
            seeds = (7 + np.arange(K)).astype(np.uint64)
            for shared, fmode in ((True, "split"), (False, "split"), (True, "exact")):
                lk.set_batched_f32_mode(fmode)
                lk.mp_reset_batched(64, blocks, K)
                for call in range(3):
                    ph = np.full(K, call % 8, dtype=np.int32) if shared else rng.integers(0, 8, K).astype(np.int32)
                    v, nb, nd = lk.mp_loglik_batched(Th + 0.001 * call, 8, ph, seeds, call, 1.7e-7)
                    h.update(v.tobytes()); h.update(nb.tobytes())
    print(h.hexdigest(), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        outs = []
        for lib in sys.argv[1:3]:
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(lib)],
                               capture_output=True, text=True, timeout=300)
            if o.returncode != 0:
                print(o.stderr[-2000:])
                raise SystemExit(1)
            outs.append(o.stdout.strip().splitlines()[-1])
            print(lib, outs[-1])
        print("IDENTICAL" if outs[0] == outs[1] else "DIFFERENT")
        raise SystemExit(0 if outs[0] == outs[1] else 2)
