import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import Likelihood
d = 64
N = int(os.environ.get("ROWS", 10000))
for timing in (False, True):
    with Likelihood(N, d, timing=timing) as lk:
        lk.fill_synthetic(1)
        rng = np.random.default_rng(0)
        inc = rng.normal(0, 1e-4, (4096, d)); logu = np.log(rng.uniform(size=4096))
        cp = lk.chain_begin(np.zeros(d), 10.0)
        lk.chain_run(inc[:300], logu[:300])
        for n in (20, 256, 2048):
            tt = []
            for rep in range(12):
                t0 = time.perf_counter()
                lk.chain_run(inc[:n], logu[:n])
                tt.append(time.perf_counter() - t0)
            print("stride=%s timing=%s N=%d n=%d total %.1f us  %.2f us/it" % (os.environ.get("MHLIK_CHAIN_STRIDE"), timing, N, n, np.median(tt) * 1e6, np.median(tt) * 1e6 / n), flush=True)
