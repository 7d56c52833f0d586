#!/usr/bin/env python3
"""Random shapes of the K-chain C samplers: the device-resident lock-step loop (chunk pipeline, helper thread) against
the per-iteration host loop (MCMC_AMD_HOST_LOOP=1) on the same generators -- samples, burn rows and accept counts bit for
bit, in double, float (both MFMA modes) and mixed precision (shared phase / phase per chain); and chain 0 of the double
run against the CPU restatement's cpu_sampler.  Run by hand on the GPU box, not collected by pytest.  python tests/fuzz/fuzz_kchain.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from mcmc_amd import Likelihood  # noqa: E402
from mcmc_amd import samplers as S  # noqa: E402
from mcmc_amd.likelihood import synthetic_host  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for c in range(cases):
        N = int(np.exp(rng.uniform(np.log(64), np.log(200000))))
        d = int(np.exp(rng.uniform(0, np.log(128))))
        if N * d > 4_000_000:
            N = max(64, 4_000_000 // d)
        K = int(rng.integers(1, 65))
        Ns, burnin = int(rng.integers(1, 300)), int(rng.integers(1, 60))
        rwsd = 2.38 / np.sqrt(d) * float(rng.choice([0.02, 0.1, 0.5]))
        seeds = [int(v) for v in rng.integers(0, 1 << 31, K)]
        kind = str(rng.choice(["f64", "f32_split", "f32_exact", "mp_shared", "mp_own"]))
        tag = "case %d N=%d d=%d K=%d Ns=%d burnin=%d rwsd=%.3g %s" % (c, N, d, K, Ns, burnin, rwsd, kind)
        X = synthetic_host(int(rng.integers(1, 1 << 30)), N, d)
        Xf = np.asfortranarray(X.astype(np.float32))
        outs = []
        with Likelihood(N, d, precision=3, segments=1 if kind.startswith("mp") else int(rng.choice([1, 8]))) as lk:
            lk.upload(X)
            lk.upload(Xf)
            if kind.startswith("f32"):
                lk.set_batched_f32_mode(kind[4:])
            for host_loop in ("0", "1"):
                os.environ["MCMC_AMD_HOST_LOOP"] = host_loop
                if kind.startswith("mp"):
                    br = int(rng.choice([64, 256])) if host_loop == "0" else br
                    o = S.mp_multichain_sampler(lk, seeds, Ns, burnin, rwsd, br, max(1, N // br), 4, 1.7e-7,
                                                shared_phase=kind == "mp_shared", theta0=np.zeros(d))
                else:
                    o = S.multichain_sampler(lk, seeds, Ns, burnin, rwsd, use_f32=kind.startswith("f32"))
                outs.append(o)
        a, b = outs
        assert np.array_equal(a["samples"], b["samples"]) and np.array_equal(a["burn"], b["burn"]) and \
            np.array_equal(a["accepted"], b["accepted"]), tag
        if kind == "f64":
            ref = O.cpu_sampler(X, seeds[0], Ns, burnin, rwsd, trace=False)
            assert np.array_equal(ref["samples"], a["samples"][0]) and np.array_equal(ref["burn"], a["burn"][0][: max(burnin, 1)]), tag
        print(tag, "accepted", int(np.sum(a["accepted"])), flush=True)
    print("fuzz_kchain: %d cases clean" % cases)


if __name__ == "__main__":
    main()
