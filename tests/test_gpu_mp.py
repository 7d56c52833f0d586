"""Mixed-precision likelihood (mcmc_mp semantics) on the MI355X vs its CPU
restatement (oracle/mp_path.c).  "Parity unpinned": the reference's MP path is
work in progress and commits no outputs."""
import ctypes as C

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


def _oracle_mp(X, Xf, th, block_rows, blocks, rf, rc, seed, call, eps, z):
    L = O.lib()
    L.oracle_mp_likelihood.restype = C.c_double
    L.oracle_mp_likelihood.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    nb, nd = C.c_int(), C.c_int()
    N, d = X.shape
    v = L.oracle_mp_likelihood(O.ptr(X), O.ptr(Xf), N, d, N, O.ptr(th), block_rows, blocks, rf, rc, seed, call, eps,
                               O.ptr(z), C.byref(nb), C.byref(nd))
    return v, nb.value, nd.value


@pytest.mark.parametrize("N,d,block_rows", [(16384, 8, 64), (10000, 3, 64), (20000, 16, 256), (8192, 5, 1024), (5000, 4, 128)])
def test_mp_likelihood_sequence_matches_oracle(N, d, block_rows):
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(99, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    blocks = N // block_rows
    rf, eps = 2, 1.7e-7
    z = np.ones(blocks, dtype=np.int8)
    rng = np.random.default_rng(0)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.upload(X)
        lk.upload(Xf)
        lk.mp_reset(block_rows, blocks)
        for call in range(6):
            th = rng.normal(0, 0.5, d)
            rc = call % rf
            want, wb, wd = _oracle_mp(X, Xf, th, block_rows, blocks, rf, rc, 1234, call, eps, z)
            got, gb, gd = lk.mp_loglik(th, block_rows, blocks, rf, rc, 1234, call, eps)
            assert (gb, gd) == (wb, wd), (call, gb, gd, wb, wd)
            assert abs(got - want) <= 2e-6 * abs(want), (call, got, want)


def test_blocks_beyond_the_data_keep_their_indicator_and_add_nothing():
    """More blocks than the matrix has rows for (the reference's mp_host.c hard-codes 16 x 64 whatever -sz is): a block
    with no rows contributes nothing and stays bright (oracle/mp_path.c), also right after a larger evaluation left
    other records in the scratch buffer."""
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d, block_rows, blocks = 5000, 4, 64, 100  # 79 blocks hold rows, 21 hold none
    X = synthetic_host(17, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    rng = np.random.default_rng(5)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.upload(X)
        lk.upload(Xf)
        lk.mp_reset(64, N // 64)  # fills the record scratch with real records first
        lk.mp_loglik(rng.normal(0, 0.5, d), 64, N // 64, 2, 0, 1234, 0, 1.7e-7)
        lk.mp_reset(block_rows, blocks)
        z = np.ones(blocks, dtype=np.int8)
        for call in range(4):
            th = rng.normal(0, 0.5, d)
            want, wb, wd = _oracle_mp(X, Xf, th, block_rows, blocks, 2, call % 2, 1234, call, 1.7e-7, z)
            got, gb, gd = lk.mp_loglik(th, block_rows, blocks, 2, call % 2, 1234, call, 1.7e-7)
            assert (gb, gd) == (wb, wd) and gb >= 21, (call, gb, gd, wb, wd)
            assert abs(got - want) <= 2e-6 * abs(want), (call, got, want)
    # (the K-chain entry point takes its indicators per (block, chain) and its sums per tile: empty blocks are covered
    # by tests/test_gpu_mp_batched.py, N=3000 with 50 blocks of 64)


def test_mp_bound_matches_oracle():
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d = 30000, 6
    X = synthetic_host(5, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    L = O.lib()
    L.oracle_mp_bound.restype = C.c_double
    L.oracle_mp_bound.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.POINTER(C.c_double)]
    th = np.random.default_rng(1).normal(0, 1.0, d)
    eps = C.c_double()
    want = L.oracle_mp_bound(O.ptr(X), O.ptr(Xf), N, d, N, O.ptr(th), C.byref(eps))
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.upload(X)
        lk.upload(Xf)
        s, e = lk.mp_bound(th)
    assert abs(s - want) <= 1e-10 * abs(want)
    assert abs(e - eps.value) <= 1e-9


def test_mp_sampler_runs():
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 4096, 3
    X = synthetic_host(12345, N, d)
    out = S.mp_sampler(X, 1, 200, 50, 0.05, block_rows=64, blocks=N // 64, resamp_factor=2, bound=1.7e-7)
    assert np.isfinite(out["samples"]).all()
    assert out["mp"].brightBlocks + out["mp"].darkBlocks == N // 64


def test_deterministic_float_sigmoid_is_bit_identical_on_host_and_device():
    """The mixed-precision path takes log(sigma_D - (sigma_S - eps)): the float
    sigmoid must be the same function in the oracle (host) and in the kernels
    (gfx950).  tools/check_detmath compares 2^20 arguments bit for bit."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_detmath")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "exp mismatches 0, sigmoid mismatches 0" in out.stdout, out.stdout
