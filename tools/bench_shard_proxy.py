#!/usr/bin/env python3
"""One rank's share of a sharded MH step, measured on ONE GPU: rows = N/G of
the N=10M, d=64 workload, with the RCCL path forced on through a one-rank
communicator (MHLIK_FORCE_COMM=1), so that segment sums -> all-reduce ->
step kernel run exactly as they do with G ranks.  What it cannot show is the
cross-GPU latency of the all-reduce itself.
  MHLIK_FORCE_COMM=1 python tools/bench_shard_proxy.py [G] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import Likelihood, F64  # noqa: E402
from mcmc_amd import samplers as S  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
N, d = 10_000_000 // G, 64
for comm in (False, True):
    with Likelihood(N, d, precision=F64, timing=True) as lk:
        lk.fill_synthetic(12345)
        if comm:
            os.environ["MHLIK_FORCE_COMM"] = "1"
            lk.comm_init(Likelihood.make_comm_id())
        with S.attached(lk):
            S.gpu_sampler(None, 0, 50, 2, 2.38 / np.sqrt(d) * 2e-3, trace=False, n_rows=N, n_cols=d)  # warm-up
            t0 = time.perf_counter()
            r = S.gpu_sampler(None, 0, steps, 2, 2.38 / np.sqrt(d) * 2e-3, trace=False, n_rows=N, n_cols=d)
            el = time.perf_counter() - t0
        print("rows/GPU=%d (G=%d) %s: %.1f us/step, kernel %.1f us -> fixed overhead %.1f us/step; "
              "8-GPU strong-scaling bound = %.2fx of the 1-GPU 729 us step" %
              (N, G, "RCCL path (1-rank comm)" if comm else "plain path", el / (steps + 2) * 1e6,
               r["res"].kernelTime * 1e3, el / (steps + 2) * 1e6 - r["res"].kernelTime * 1e3,
               729.0 / (el / (steps + 2) * 1e6)), flush=True)
