#!/usr/bin/env python3
"""A/B timing of the K-chain passes on ONE box: each library given on the command line is loaded in a child
process and timed in turn, several rounds, so that box-to-box and warm-up differences cancel.
usage: python tools/ab_batched.py libA.so libB.so [rounds]      (child: --child lib)"""
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
    N, d, K = 10_000_000, 128, 64
    rng = np.random.default_rng(0)
    Th = rng.normal(0, 0.01, (K, d))
    out = []
    with Likelihood(N, d, precision=3, segments=1, timing=True) as lk:
        lk.fill_synthetic(12345)
        for name, setup, dt in (("x3", lambda: lk.set_batched_f32_mode("split"), np.float32),
                                ("f32", lambda: lk.set_batched_f32_mode("exact"), np.float32),
                                ("f64", lambda: None, np.float64)):
            setup()
            ms = []
            for r in range(14):
                lk.loglik_batched(Th.astype(dt), dtype=dt)
                ms.append(lk.last_kernel()[0])
            out.append("%s %.4f" % (name, float(np.median(ms[4:]))))
    print(" ".join(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        libs = [os.path.abspath(p) for p in sys.argv[1:3]]
        rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
        for r in range(rounds):
            for lib in libs:
                res = subprocess.run([sys.executable, __file__, "--child", lib], capture_output=True, text=True, timeout=300)
                print(os.path.basename(lib), res.stdout.strip() or res.stderr[-300:], flush=True)
