#!/usr/bin/env python3
"""Where the fixed cost of a short device-chain run goes: wall time of resume + submit + collect + state for n
iterations at a small N (kernels of a few microseconds, so the set-up shows) and at N=10M."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import Likelihood

def main():
    d = 64
    for N in (10_000, 10_000_000):
        with Likelihood(N, d, timing=True) as lk:
            lk.fill_synthetic(1)
            rng = np.random.default_rng(0)
            inc = rng.normal(0, 1e-4, (4096, d)); logu = np.log(rng.uniform(size=4096))
            cp = lk.chain_begin(np.zeros(d), 10.0)
            lk.chain_run(inc[:300], logu[:300])
            for timing_label in ("run",):
                for n in (1, 2, 8, 20, 64, 256):
                    ts, tt, km = [], [], []
                    for rep in range(12):
                        t0 = time.perf_counter()
                        lk.chain_resume(np.zeros(d), cp, 10.0)
                        lk.chain_submit(inc[:n], logu[:n])
                        t1 = time.perf_counter()
                        o = lk.chain_collect(n)
                        lk.chain_state()
                        t2 = time.perf_counter()
                        ts.append(t1 - t0); tt.append(t2 - t0); km.append(o["kernel_ms"])
                    ts, tt, km = np.median(ts) * 1e6, np.median(tt) * 1e6, np.median(km) * 1e3
                    print("N=%d n=%d submit-returns %.1f us total %.1f us  kernel %.2f us  total-n*kernel %.1f us (%.2f us/it)"
                          % (N, n, ts, tt, km, tt - n * km, (tt - n * km) / n), flush=True)
main()
