#!/usr/bin/env python3
"""Random shapes of the DP sampler on the MI355X against the CPU restatement's cpu_sampler with the same seed: the
accept / reject sequence, every sample and burn-in row bit for bit, likelihood trace within 1e-10 (development tool, GPU
box).  A run whose smallest decision margin |log u - ratio| is below 1e-8 is reported, not failed, if it diverges there.
python tests/fuzz/fuzz_sampler.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from mcmc_amd import samplers as S  # noqa: E402
from mcmc_amd.likelihood import synthetic_host  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    close = 0
    for c in range(cases):
        N = int(np.exp(rng.uniform(0, np.log(150000))))
        d = int(np.exp(rng.uniform(0, np.log(100))))
        if N * d > 3_000_000:
            N = max(1, 3_000_000 // d)
        Ns, burnin = int(rng.integers(1, 400)), int(rng.integers(1, 120))
        rwsd = 2.38 / np.sqrt(d) * float(rng.choice([0.01, 0.05, 0.3, 1.0]))
        seed = int(rng.integers(0, 1 << 31))
        os.environ["MHLIK_CHAIN_PERSIST"] = str(rng.integers(0, 2))
        X = synthetic_host(int(rng.integers(1, 1 << 30)), N, d)
        ref = O.cpu_sampler(X, seed, Ns, burnin, rwsd)
        got = S.gpu_sampler(X, seed, Ns, burnin, rwsd)
        tag = "case %d N=%d d=%d Ns=%d burnin=%d rwsd=%.3g seed=%d persist=%s acceptance %.3f" % (
            c, N, d, Ns, burnin, rwsd, seed, os.environ["MHLIK_CHAIN_PERSIST"], ref["acceptance"])
        margin = np.abs(ref["trace"]["logu"] - ref["trace"]["ratio"]).min() if len(ref["trace"]) else np.inf
        same = (np.array_equal(ref["trace"]["accept"], got["trace"]["accept"]) and np.array_equal(ref["samples"], got["samples"])
                and np.array_equal(ref["burn"], got["burn"]) and ref["acceptance"] == got["acceptance"])
        if not same and margin < 1e-8:
            close += 1
            print(tag, "-- diverges at a decision margin of %.3g: not counted" % margin, flush=True)
            continue
        assert same, tag
        if len(ref["trace"]):
            rel = np.abs(ref["trace"]["lik"] - got["trace"]["lik"]) / np.maximum(np.abs(ref["trace"]["lik"]), 1e-300)
            assert rel.max() <= 1e-10, (tag, rel.max())
        print(tag, flush=True)
    print("fuzz_sampler: %d cases clean, %d too close to call" % (cases - close, close))


if __name__ == "__main__":
    main()
