"""bench.py's two drivers of the MH step -- the C sampler loop (device-resident
chain) and the Python fallback loop used when the in-library RCCL communicator
cannot be set up -- walk the same chain: same seed, same accept count."""
import argparse
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fallback_loop_walks_the_same_chain_as_the_c_sampler():
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from mcmc_amd import Likelihood, F64
    from mcmc_amd import samplers as S

    N, d, K, W = 200_000, 16, 150, 10
    a = argparse.Namespace(seed=3)
    rwsd = 2.38 / np.sqrt(d) * 0.02

    def barrier():
        torch.cuda.synchronize()

    with Likelihood(N, d, precision=F64, timing=True) as lk:
        lk.fill_synthetic(12345)
        _, k1, acc_c = bench.run_c_sampler(a, lk, S, np, N, d, K, W, rwsd, barrier)
        _, k2, acc_py = bench.run_host_loop(a, lk, S, np, torch, None, torch.device("cuda", 0), 1, d, K, W, rwsd, barrier)
    assert acc_c == acc_py and 0 < acc_c < K
    assert k1 > 0 and k2 > 0
