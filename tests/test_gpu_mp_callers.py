"""The exported callers around the mixed-precision likelihood -- mp_sampler,
get_lowerbound, get_blockSz (src/mcmc_mp/mcmc_mp.cu:283-473, 652-744, 824-995)
-- and the tuners no other test reaches (tune_target_a_gpu, tune_ess_sp) on
the MI355X, against the oracle's restatement of the same loops
(oracle/mp_path.c).  "Parity unpinned": the reference's MP path is work in
progress and commits no outputs; what is checked is our statement of it,
iteration by iteration."""
import ctypes as C

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu

MP_REL = 2e-6  # float-accumulated dark sums: summation order differs between the CPU loop and the GPU tree


def _mp_data(seed, N, d):
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(seed, N, d)
    return X, np.asfortranarray(X.astype(np.float32))


def _oracle():
    L = O.lib()
    P = C.c_void_p
    L.oracle_get_lowerbound.restype = C.c_double
    L.oracle_get_lowerbound.argtypes = [P, P, C.c_long, C.c_int, C.c_long, P, C.c_double, C.c_int, P, C.c_long,
                                        C.POINTER(C.c_long)]
    L.oracle_get_blockSz_runs.restype = C.c_int
    L.oracle_get_blockSz_runs.argtypes = [P, P, C.c_long, C.c_int, C.c_long, P, C.c_double, C.c_double, P, C.c_int, P,
                                          C.c_long, C.POINTER(C.c_long)]
    L.oracle_mp_sampler.restype = None
    L.oracle_mp_sampler.argtypes = [P, P, C.c_long, C.c_int, C.c_long, P, C.c_double, C.c_int, C.c_int, C.c_int,
                                    C.c_double, C.c_int, C.c_int, P, P, C.POINTER(C.c_int), C.POINTER(C.c_int), P,
                                    C.c_long, C.POINTER(C.c_long)]
    return L


def _compare_traces(want, got, what):
    assert len(want) == len(got), (what, len(want), len(got))
    assert np.array_equal(want["accept"], got["accept"]), "%s: accept/reject sequence differs" % what
    assert np.array_equal(want["logu"], got["logu"]), "%s: the uniforms differ (RNG stream out of step)" % what
    rel = np.abs(want["lik"] - got["lik"]) / np.maximum(np.abs(want["lik"]), 1e-300)
    assert rel.max() <= MP_REL, (what, rel.max())


@pytest.mark.parametrize("N,d,block_rows,rf,seed", [(8192, 3, 64, 2, 1), (16384, 5, 128, 16, 2), (20000, 4, 256, 10, 3)])
def test_mp_sampler_chain_matches_oracle(N, d, block_rows, rf, seed):
    from mcmc_amd import samplers as S

    X, Xf = _mp_data(12345, N, d)
    blocks, Ns, burnin, sd, eps = N // block_rows, 150, 60, 0.04, 1.7e-7
    L = _oracle()
    samples, burn = np.zeros((Ns, d)), np.zeros((burnin, d))
    tr = np.zeros(Ns + burnin + 8, dtype=O.TRACE_DTYPE)
    acc, bright, ntr = C.c_int(), C.c_int(), C.c_long()
    rng = O.Rng(seed, L)
    L.oracle_mp_sampler(O.ptr(X), O.ptr(Xf), N, d, N, rng.r, sd, block_rows, blocks, rf, eps, burnin, Ns, O.ptr(burn),
                        O.ptr(samples), C.byref(acc), C.byref(bright), O.ptr(tr), len(tr), C.byref(ntr))
    got = S.mp_sampler(X, seed, Ns, burnin, sd, block_rows, blocks, rf, eps)
    _compare_traces(tr[:ntr.value], got["trace"], "mp_sampler")
    assert np.array_equal(samples, got["samples"])
    assert np.array_equal(burn[1:], got["burn"][1:])  # row 0 is never written (the chain starts at theta = 1)
    assert got["acceptance"] == float(np.float32(acc.value / Ns))
    assert got["mp"].brightBlocks == bright.value and got["mp"].darkBlocks == blocks - bright.value
    assert 0 < acc.value < Ns


def test_get_lowerbound_matches_oracle():
    """50 000 iterations of the lower-bound search (LOWERBOUND_REPS, mcmc_mp.cu:23) on a small matrix: the same
    accept/reject sequence, the same sums of sigmoids, the same bound."""
    from mcmc_amd import samplers as S

    N, d, seed, sd = 1000, 3, 4, 0.3
    X, Xf = _mp_data(77, N, d)
    L = _oracle()
    cap = 50_000
    tr = np.zeros(cap, dtype=O.TRACE_DTYPE)
    ntr = C.c_long()
    orng = O.Rng(seed, L)  # kept alive: the handle is freed with the object
    want = L.oracle_get_lowerbound(O.ptr(X), O.ptr(Xf), N, d, N, orng.r, sd, 50_000, O.ptr(tr), cap, C.byref(ntr))
    lib = S.load_samplers()
    gtr = np.zeros(cap, dtype=S.TRACE_DTYPE)
    init = np.zeros(d)
    rng = S.Rng(seed)
    lib.mcmc_amd_trace_begin(S._ptr(gtr), cap)
    try:
        got = lib.get_lowerbound(S.data_str(data=S._ptr(X).value, dataf=S._ptr(Xf).value), S._ptr(init), rng.r,
                                 S.mcmc_str(ddata=d, Nd=N, Ns=1, burnin=1, impl=S.IMPL_MP),
                                 S.mcmc_tune_str(rwsd=sd, rwsdf=sd))
    finally:
        n = lib.mcmc_amd_trace_end()
    assert n == ntr.value == 49_999
    want_tr, got_tr = tr[:n], gtr[:n]
    assert np.array_equal(want_tr["accept"], got_tr["accept"])
    assert np.array_equal(want_tr["logu"], got_tr["logu"])
    assert (np.abs(want_tr["lik"] - got_tr["lik"]) <= 1e-10 * np.abs(want_tr["lik"])).all()
    # epsilon is a difference of a double and a float sigmoid (~1e-8): the device exp() may differ from glibc's by an ulp.
    # (The search starts at theta = 0, where both sigmoids are exactly 0.5 and the extreme is 0; rows with a negative
    # difference exist for every other theta, so 0 is what this definition returns -- the reference takes abs() of it,
    # mp_host.c:53, and then hard-codes a constant instead.)
    assert abs(got - want) <= 1e-9 and got >= 0


def test_get_blockSz_runs_match_oracle():
    """get_blockSz picks the fastest configuration by wall time (not comparable); the configurations it tries and
    every iteration of every 500-step short run are."""
    from mcmc_amd import samplers as S

    N, d, seed, sd, eps = 16384, 3, 6, 0.05, 1.7e-5
    X, Xf = _mp_data(31, N, d)
    L = _oracle()
    cap = 4000
    cfg = np.zeros(3 * 16, dtype=np.int32)
    tr = np.zeros(cap, dtype=O.TRACE_DTYPE)
    ntr = C.c_long()
    orng = O.Rng(seed, L)
    ncfg = L.oracle_get_blockSz_runs(O.ptr(X), O.ptr(Xf), N, d, N, orng.r, sd, eps, O.ptr(cfg), 16, O.ptr(tr), cap,
                                     C.byref(ntr))
    tried = cfg[:3 * ncfg].reshape(ncfg, 3)
    assert [tuple(t) for t in tried] == [(64, 256, 25), (128, 128, 12)]  # blocks/10 >= 10 (mcmc_mp.cu:877)
    lib = S.load_samplers()
    gtr = np.zeros(cap, dtype=S.TRACE_DTYPE)
    mp = S.mp_str()
    init = np.zeros(d)
    rng = S.Rng(seed)
    lib.mcmc_amd_trace_begin(S._ptr(gtr), cap)
    try:
        lib.get_blockSz(S.data_str(data=S._ptr(X).value, dataf=S._ptr(Xf).value), rng.r, C.byref(mp),
                        S.mcmc_str(ddata=d, Nd=N, Ns=1, burnin=1, impl=S.IMPL_MP), S.mcmc_tune_str(rwsd=sd, rwsdf=sd),
                        S._ptr(init), eps)
    finally:
        n = lib.mcmc_amd_trace_end()
    assert n == ntr.value == ncfg * 499
    _compare_traces(tr[:n], gtr[:n], "get_blockSz short runs")
    assert (mp.dataBlockSz, mp.dataBlocks, mp.ResampFactor) in [tuple(t) for t in tried]
    # both generators are now at the same point of the stream
    lib.gsl_rng_get.restype = C.c_ulong
    lib.gsl_rng_get.argtypes = [C.c_void_p]
    assert lib.gsl_rng_get(rng.r) == L.gsl_rng_get(orng.r)


def test_tune_target_a_gpu_legacy_stepper_matches_oracle():
    """tune_target_a_gpu (mcmc_gpu.cu:477-561), the legacy +-10 % stepper, against the oracle's tune_target_a_cpu
    (mcmc_cpu.c:274-324): the same trial chains, the same selected sd."""
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d, seed = 1500, 3, 9
    X = synthetic_host(12345, N, d)
    L = O.lib()
    mvout = np.empty(N)
    mct = O.mcmc_tune_str(rwsd=2.38 / np.sqrt(d), rwsdf=0)
    init = np.zeros(d)
    orng = O.Rng(seed, L)
    L.tune_target_a_cpu(O.data_str(data=O.ptr(X).value, mvout=O.ptr(mvout).value), orng.r,
                        O.mcmc_str(ddata=d, Nd=N, Ns=1, burnin=40, impl=0), C.byref(mct), O.ptr(init))
    lib = S.load_samplers()
    gmct = S.mcmc_tune_str(rwsd=2.38 / np.sqrt(d), rwsdf=0)
    rng = S.Rng(seed)
    lib.tune_target_a_gpu(S.data_str(data=S._ptr(X).value), rng.r, S.mcmc_str(ddata=d, Nd=N, Ns=1, burnin=40, impl=1),
                          C.byref(gmct), S.gpu_v_str(), S._ptr(init), 5000)
    assert gmct.rwsd == mct.rwsd and gmct.rwsd > 0 and gmct.rwsd != 2.38 / np.sqrt(d)


def test_tune_ess_sp_is_deterministic_and_sane():
    """tune_ess_sp (mcmc_gpu_sp.cu:267-386).  No float oracle chain exists (in float the summation order decides
    accept/reject, and the reference's own order is cuBLAS's): checked for determinism, for a result inside the
    search's reach, and against the double-precision tuner's choice on the same data."""
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d, seed = 3000, 3, 2
    X = synthetic_host(12345, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    lib = S.load_samplers()
    sd0 = 2.38 / np.sqrt(d)

    def run():
        mct = S.mcmc_tune_str(rwsd=sd0, rwsdf=sd0)
        init = np.zeros(d, dtype=np.float32)
        rng = S.Rng(seed)
        lib.tune_ess_sp(S.data_str(dataf=S._ptr(Xf).value), rng.r,
                        S.mcmc_str(ddata=d, Nd=N, Ns=1, burnin=30, impl=S.IMPL_SP), C.byref(mct), S.gpu_v_str(),
                        S._ptr(init), 600)  # chain length >= the fixed 500 lags (mcmc_gpu_sp.cu:288)
        return float(mct.rwsdf)

    a, b = run(), run()
    assert a == b and np.isfinite(a) and a > 0
    dp = S.tune_ess_gpu(X, seed, 30, sd0, 600)
    assert 0.2 < a / dp < 5.0, (a, dp)
    sp = S.sp_sampler(Xf, 5, 1500, 200, a, trace=False)
    assert 0.02 < sp["acceptance"] < 0.8
