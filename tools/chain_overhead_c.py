#!/usr/bin/env python3
"""Fixed cost of a short metropolis_gpu call as bench.py times it (barrier, ctypes call, C sampler, library)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcmc_amd import Likelihood, F64
from mcmc_amd import samplers as S

def main():
    N, d = int(os.environ.get("ROWS", 10_000_000)), 64
    lib = S.load_samplers()
    P = C.c_void_p
    lib.burn_in_metropolis_gpu.argtypes = [P, P, S.mcmc_str, S.mcmc_tune_str, S.mcmc_v_str, S.mcmc_int_v, P]
    lib.metropolis_gpu.argtypes = [P, P, S.mcmc_str, S.mcmc_tune_str, S.mcmc_v_str, S.mcmc_int_v, P,
                                   C.POINTER(C.c_int), C.POINTER(S.out_str)]
    with Likelihood(N, d, precision=F64, timing=True) as lk:
        lk.fill_synthetic(12345)
        rwsd = 2.38 / np.sqrt(d) * 2e-3
        for K in (20, 100, 300, 300):
            samples = np.zeros((K, d)); burn = np.zeros((5, d))
            mcin = S.mcmc_str(ddata=d, Nd=N, Ns=K, burnin=5, impl=S.IMPL_GPU)
            mct = S.mcmc_tune_str(rwsd=rwsd, rwsdf=rwsd)
            mcdata = S.mcmc_v_str(samples=samples.ctypes.data, burn=burn.ctypes.data)
            prop, cur = np.zeros(d), np.zeros(d)
            mclocv = S.mcmc_int_v(proposed=prop.ctypes.data, current=cur.ctypes.data)
            mcloc = S.mcmc_int(); res = S.out_str(); accepted = C.c_int(0)
            rng = S.Rng(0)
            lib.burn_in_metropolis_gpu(lk.handle, rng.r, mcin, mct, mcdata, mclocv, C.byref(mcloc))
            el = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                lib.metropolis_gpu(lk.handle, rng.r, mcin, mct, mcdata, mclocv, C.byref(mcloc), C.byref(accepted), C.byref(res))
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                el.append((t1 - t0, t2 - t0, float(res.kernelTime)))
            for a, b, k in el:
                print("N=%d K=%d call %.1f us  +sync %.1f us  kernel %.2f us  call-K*kernel %.1f us" % (N, K, a * 1e6, b * 1e6, k * 1e3, a * 1e6 - K * k * 1e3), flush=True)
        t0 = time.perf_counter()
        for _ in range(100): torch.cuda.synchronize()
        print("torch.cuda.synchronize: %.1f us" % ((time.perf_counter() - t0) * 1e4))
main()
