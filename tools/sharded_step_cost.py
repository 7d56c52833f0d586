#!/usr/bin/env python3
"""What one MH iteration costs beside the likelihood kernel when the chain runs in its multi-GPU form, measured on
ONE GPU: a context of N/8 rows (a rank's share of BASELINE configs[1] on 8 GPUs) with a one-rank RCCL communicator
(the in-stream ncclAllReduce, the slot kernels and the step kernel are all on the path; only the peers are missing)
against the same context without a communicator.  Development tool: python tools/sharded_step_cost.py [rows] [steps]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    import torch

    from mcmc_amd import Likelihood, F64
    from mcmc_amd import samplers as S

    a = argparse.Namespace(seed=0)
    d = 64
    rwsd = 2.38 / np.sqrt(d) * 2e-3

    def barrier():
        torch.cuda.synchronize()

    for with_comm in (False, True, False, True):
        with Likelihood(rows, d, precision=F64, device=0, rank=0, world=1, timing=True) as lk:
            lk.fill_synthetic(12345)
            if with_comm:
                os.environ["MHLIK_FORCE_COMM"] = "1"  # a one-rank communicator is skipped otherwise
                lk.comm_init(Likelihood.make_comm_id())
                assert lk.comm_ranks() == 1, lk.comm_ranks()
            el, kms, acc = bench.run_c_sampler(a, lk, S, np, rows, d, steps, 300, rwsd, barrier)
            print("rows %d  comm_ranks %d  %-28s %8.1f it/s  %.2f us per iteration, kernel %.2f us, beside the kernel %.2f us  (%s)" %
                  (rows, lk.comm_ranks(), "one-rank RCCL communicator" if with_comm else "no communicator", steps / el, el / steps * 1e6,
                   kms * 1e3, el / steps * 1e6 - kms * 1e3, lk.last_kernel_name()), flush=True)


if __name__ == "__main__":
    main()
