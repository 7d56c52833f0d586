"""K lock-step MH chains kept on the device (mhlik_kchain_*, SURVEY.md section
8 row f3 for K chains; no counterpart in the reference, whose loop is
src/mcmc_gpu/mcmc_gpu.cu:101-128 for one chain): the device-resident loop walks
the chains the per-iteration host loop of samplers/multichain.c walks -- which
tests/test_gpu_golden.py and tests/test_gpu_batched.py tie to the oracle -- bit
for bit, in every mode, and the raw entry points agree with the batched
likelihood they are built on."""
import os

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


class _host_loop:
    """MCMC_AMD_HOST_LOOP=1 for the duration of a block (read per call by the C samplers)."""

    def __enter__(self):
        self.old = os.environ.get("MCMC_AMD_HOST_LOOP")
        os.environ["MCMC_AMD_HOST_LOOP"] = "1"

    def __exit__(self, *exc):
        if self.old is None:
            del os.environ["MCMC_AMD_HOST_LOOP"]
        else:
            os.environ["MCMC_AMD_HOST_LOOP"] = self.old


@pytest.mark.parametrize("use_f32,f32_mode", [(False, None), (True, "exact"), (True, "split")])
@pytest.mark.parametrize("N,d,K,Ns,burnin", [(30000, 8, 5, 700, 40), (9000, 37, 64, 60, 1), (20011, 16, 33, 50, 0)])
def test_device_loop_equals_host_loop(N, d, K, Ns, burnin, use_f32, f32_mode):
    from mcmc_amd import Likelihood, samplers as S

    seeds = list(range(3, 3 + K))
    rwsd = 2.38 / np.sqrt(d) * 0.02
    with Likelihood(N, d, precision=2 if use_f32 else 1) as lk:
        lk.fill_synthetic(99)
        if f32_mode:
            lk.set_batched_f32_mode(f32_mode)
        dev = S.multichain_sampler(lk, seeds, Ns, burnin, rwsd, use_f32=use_f32)
        with _host_loop():
            host = S.multichain_sampler(lk, seeds, Ns, burnin, rwsd, use_f32=use_f32)
    assert np.array_equal(dev["samples"], host["samples"])
    assert np.array_equal(dev["burn"], host["burn"])
    assert np.array_equal(dev["accepted"], host["accepted"])
    if burnin > 0:  # without a burn-in there is no chain start: cposterior stays 0 and nothing is ever accepted
        assert 0 < dev["accepted"].sum() < K * Ns


def test_device_loop_chain_equals_cpu_sampler():
    """Chain k of the device-resident K-chain loop is cpu_sampler with generator k: accept/reject sequence and samples
    identical to the oracle's (the double batched pass is within 1e-10 of the oracle's likelihood)."""
    from mcmc_amd import Likelihood, samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d, K, Ns, burnin = 20000, 6, 4, 400, 30
    X = synthetic_host(12345, N, d)
    rwsd = 0.03
    with Likelihood(N, d) as lk:
        lk.upload(X)
        got = S.multichain_sampler(lk, [10, 11, 12, 13], Ns, burnin, rwsd)
    for k, seed in enumerate([10, 11, 12, 13]):
        ref = O.cpu_sampler(X, seed, Ns, burnin, rwsd)
        assert np.array_equal(ref["samples"], got["samples"][k]), k
        assert np.array_equal(ref["burn"][1:], got["burn"][k][1:]), k


def test_raw_entry_points_against_the_batched_likelihood():
    """mhlik_kchain_begin / _run / _state: the likelihoods the step kernel records are the batched pass's, the ratios
    and decisions follow from them, the state read back is the last sample row."""
    from mcmc_amd import Likelihood

    N, d, K, n = 50000, 12, 7, 9
    rng = np.random.default_rng(4)
    theta0 = rng.normal(0, 0.1, (K, d))
    inc = rng.normal(0, 0.004, (n, K, d))
    logu = np.log(rng.uniform(size=(n, K)))
    with Likelihood(N, d) as lk:
        lk.fill_synthetic(5)
        cp = lk.kchain_begin(theta0, 10.0)
        L0 = lk.loglik_batched(theta0)
        prior0 = np.array([np.sum(np.log(np.exp(-(t / 10.0) ** 2 / 2) / (np.sqrt(2 * np.pi) * 10.0))) for t in theta0])
        assert np.allclose(cp, prior0 + L0, rtol=1e-13, atol=0)
        out = lk.kchain_run(inc, logu)
        cur = theta0.copy()
        cpost = cp.copy()
        for it in range(n):
            prop = cur + inc[it]
            L = lk.loglik_batched(prop)
            assert np.array_equal(L, out["lik"][it]), it  # the same kernels, the same bits
            for k in range(K):
                ratio = out["ratio"][it, k]
                assert (logu[it, k] <= ratio) == bool(out["accepted"][it, k])
                if out["accepted"][it, k]:
                    cur[k] = prop[k]
                    cpost[k] = cpost[k] + ratio
            assert np.array_equal(out["samples"][it], cur)
        state, cp_end = lk.kchain_state(K)
        assert np.array_equal(state, cur)
        assert np.allclose(cp_end, cpost, rtol=1e-12, atol=0)
        # a second batch continues from the device-side state
        out2 = lk.kchain_run(inc[:2], logu[:2])
        assert np.array_equal(out2["lik"][0], lk.loglik_batched(cur + inc[0]))


def test_kchain_rejects_bad_arguments():
    from mcmc_amd import Likelihood
    from mcmc_amd._lib import MhlikError

    with Likelihood(1000, 4) as lk:
        lk.fill_synthetic(1)
        with pytest.raises(MhlikError, match="no K-chain state"):
            lk.kchain_run(np.zeros((1, 2, 4)), np.zeros((1, 2)))
        with pytest.raises(MhlikError, match="outside"):
            lk.kchain_begin(np.zeros((65, 4)), 10.0)
        with pytest.raises(MhlikError, match="not on the device"):
            lk.kchain_begin(np.zeros((2, 4)), 10.0, mode=2)
        lk.kchain_begin(np.zeros((2, 4)), 10.0)
        out = lk.kchain_run(np.zeros((0, 2, 4)), np.zeros((0, 2)))
        assert out["samples"].shape == (0, 2, 4)


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("f32_mode,segments,force", [("exact", 1, False), ("split", 1, False), ("split", 8, False),
                                                     ("exact", 8, True)])
def test_mp_device_loop_equals_host_loop(shared, f32_mode, segments, force, monkeypatch):
    """The mixed-precision K-chain sampler (mcmc_mp.cu:368-473 per chain) with the loop on the device against the
    per-iteration host loop: same chains, same indicators at the end -- on one segment, on eight block-aligned
    segments (the per-segment slot path) and over a forced one-rank RCCL communicator (the path N ranks take)."""
    from mcmc_amd import Likelihood, samplers as S

    N, d, K, Ns, burnin = 16384, 6, 20, 90, 25
    seeds = list(range(50, 50 + K))
    th0 = np.full(d, 0.05)
    if force:
        monkeypatch.setenv("MHLIK_FORCE_COMM", "1")
    with Likelihood(N, d, precision=3, segments=segments, segment_align=1024 if segments > 1 else 0) as lk:
        if force:
            lk.comm_init(Likelihood.make_comm_id())
        lk.fill_synthetic(7)
        lk.set_batched_f32_mode(f32_mode)
        lk._mpb = (64, N // 64, K)  # geometry the C sampler resets the context to (for mp_indicators_batched)
        dev = S.mp_multichain_sampler(lk, seeds, Ns, burnin, 0.01, 64, N // 64, 8, 1.7e-7, shared_phase=shared, theta0=th0)
        z_dev = lk.mp_indicators_batched()
        with _host_loop():
            host = S.mp_multichain_sampler(lk, seeds, Ns, burnin, 0.01, 64, N // 64, 8, 1.7e-7, shared_phase=shared,
                                           theta0=th0)
        z_host = lk.mp_indicators_batched()
    assert np.array_equal(dev["samples"], host["samples"])
    assert np.array_equal(dev["burn"][:, 1:], host["burn"][:, 1:])
    assert np.array_equal(dev["accepted"], host["accepted"])
    assert np.array_equal(dev["bright"], host["bright"])
    assert np.array_equal(z_dev, z_host)
    assert 0 < dev["accepted"].sum() < K * Ns


@pytest.mark.parametrize("mode,dtype", [(1, np.float64), (2, np.float32)])
def test_kchain_over_forced_rccl_is_bitwise_equal(monkeypatch, mode, dtype):
    """The K-chain loop as N ranks run it -- separate segment-sum, in-stream ncclAllReduce of the K x segments slots,
    step kernel, Theta image kernel -- on a one-rank communicator (MHLIK_FORCE_COMM=1), against the fused
    single-GPU loop: same likelihoods, decisions and samples, bit for bit."""
    from mcmc_amd import Likelihood

    N, d, K, n = 60_000, 12, 9, 40
    rng = np.random.default_rng(8)
    theta0 = rng.normal(0, 0.05, (K, d))
    inc = rng.normal(0, 0.004, (n, K, d))
    logu = np.log(rng.uniform(size=(n, K)))

    def run(force):
        with Likelihood(N, d, precision=3, device=0) as lk:
            lk.fill_synthetic(17)
            if force:
                lk.comm_init(Likelihood.make_comm_id())
                assert lk.comm_ranks() == 1
            cp = lk.kchain_begin(theta0, 10.0, mode=mode)
            out = lk.kchain_run(inc, logu)
            state, cpe = lk.kchain_state(K)
            return cp, out, state, cpe

    cp0, plain, st0, ce0 = run(False)
    monkeypatch.setenv("MHLIK_FORCE_COMM", "1")
    cp1, forced, st1, ce1 = run(True)
    assert np.array_equal(cp0, cp1)
    for k in ("samples", "lik", "ratio", "accepted"):
        assert np.array_equal(plain[k], forced[k]), k
    assert np.array_equal(st0, st1) and np.array_equal(ce0, ce1)
    assert 0 < plain["accepted"].sum() < n * K


@pytest.mark.parametrize("use_f32", [False, True])
def test_multichain_sampler_over_forced_rccl(monkeypatch, use_f32):
    """The C K-chain sampler on a context with a (one-rank) RCCL communicator -- what every rank of a sharded job
    runs -- walks the chains of the plain single-GPU run, device loop and host loop alike."""
    from mcmc_amd import Likelihood, samplers as S

    N, d, K, Ns, burnin = 40_000, 9, 11, 120, 30
    seeds = list(range(20, 20 + K))
    rwsd = 2.38 / np.sqrt(d) * 0.03

    def run(force, host):
        with Likelihood(N, d, precision=2 if use_f32 else 1, device=0) as lk:
            lk.fill_synthetic(4)
            if force:
                lk.comm_init(Likelihood.make_comm_id())
            if host:
                with _host_loop():
                    return S.multichain_sampler(lk, seeds, Ns, burnin, rwsd, use_f32=use_f32)
            return S.multichain_sampler(lk, seeds, Ns, burnin, rwsd, use_f32=use_f32)

    plain = run(False, False)
    monkeypatch.setenv("MHLIK_FORCE_COMM", "1")
    for host in (False, True):
        got = run(True, host)
        assert np.array_equal(plain["samples"], got["samples"])
        assert np.array_equal(plain["burn"], got["burn"])
        assert np.array_equal(plain["accepted"], got["accepted"])
