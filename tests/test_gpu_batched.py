"""K proposals per pass (X.Theta on MFMA) vs the oracle's K separate
log_likelihood calls.  No reference counterpart (BASELINE config 5)."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


def _oracle_batched(X, Theta):
    return np.array([O.log_likelihood(X, th)[0] for th in Theta])


@pytest.mark.parametrize("N,d,K", [(4096, 8, 64), (10000, 3, 5), (5000, 16, 33), (3000, 128, 64), (1025, 7, 1),
                                   (20000, 64, 16), (513, 1, 64), (70001, 10, 48)])
@pytest.mark.parametrize("f64_mode", ["fast", "exact"])
def test_batched_f64_matches_oracle(N, d, K, f64_mode):
    """Both epilogues of the double pass (table-driven softplus, the default, and the full-length series) against K
    oracle evaluations, to the DP bar of 1e-10 on the sum; the two agree with each other to 1e-13."""
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(12345, N, d)
    Theta = np.random.default_rng(K).normal(0, 0.3, (K, d))
    want = _oracle_batched(X, Theta)
    with Likelihood(N, d) as lk:
        lk.upload(X)
        lk.set_batched_f64_mode(f64_mode)
        got = lk.loglik_batched(Theta)
        single = np.array([lk.loglik(th) for th in Theta[:4]])
        lk.set_batched_f64_mode("exact" if f64_mode == "fast" else "fast")
        other = lk.loglik_batched(Theta)
    assert np.all(np.abs(got - want) <= 1e-10 * np.abs(want)), np.abs(got - want).max()
    assert np.all(np.abs(got[:4] - single) <= 1e-10 * np.abs(single))
    assert np.all(np.abs(got - other) <= 1e-13 * np.abs(want)), (np.abs(got - other) / np.abs(want)).max()


def test_batched_f64_fast_mode_extremes_and_full_size_sum():
    """The table-driven epilogue where its table ends: dots far beyond +-37.65 (exactly 0 / x like the reference's
    formula), overflow above 709.78 (-inf), and a sum over 2e6 rows x 64 chains against the exact mode (no drift)."""
    from mcmc_amd import Likelihood

    N, d = 4000, 2
    X = np.zeros((N, d), order="F")
    X[:, 0] = 1.0
    X[:, 1] = np.linspace(-1, 1, N)
    Theta = np.array([[50.0, 30.0], [-50.0, 30.0], [800.0, 0.0], [-800.0, 0.0], [0.0, 0.0], [37.6, 0.1], [-37.7, 0.05]])
    want = _oracle_batched(X, Theta)
    with Likelihood(N, d) as lk:
        lk.upload(X)
        got = lk.loglik_batched(Theta)
    for g, w in zip(got, want):
        if np.isinf(w):
            assert g == w
        else:
            assert abs(g - w) <= 1e-10 * abs(w) + 1e-12, (g, w)  # sums of terms below 1e-16 each: absolute
    N, d, K = 2_000_000, 16, 64
    Theta = np.random.default_rng(1).normal(0, 0.5, (K, d))
    with Likelihood(N, d) as lk:
        lk.fill_synthetic(3)
        fast = lk.loglik_batched(Theta)
        lk.set_batched_f64_mode("exact")
        exact = lk.loglik_batched(Theta)
    assert np.max(np.abs(fast - exact) / np.abs(exact)) <= 1e-14


@pytest.mark.parametrize("mode", ["exact", "split"])
@pytest.mark.parametrize("N,d,K", [(4096, 8, 64), (10000, 3, 5), (5000, 16, 33), (3000, 128, 64), (1025, 7, 1),
                                   (20000, 64, 16), (70001, 11, 48), (2500, 33, 17), (1500, 1, 2), (300000, 32, 64)])
def test_batched_f32_within_sp_tolerance(N, d, K, mode):
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(12345, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    Theta = np.random.default_rng(K).normal(0, 0.3, (K, d))
    want = _oracle_batched(X, Theta)
    with Likelihood(N, d, precision=2) as lk:
        lk.upload(Xf)
        lk.set_batched_f32_mode(mode)
        got = lk.loglik_batched(Theta.astype(np.float32), dtype=np.float32)
        single = np.array([lk.loglik_f32(th.astype(np.float32)) for th in Theta[:4]])
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want)), (np.abs(got - want) / np.abs(want)).max()
    # same float row dots as the single-theta kernel; only the exp/log flavour differs
    assert np.all(np.abs(got[:4] - single) <= 2e-6 * np.abs(single))


def test_batched_f32_split_is_as_close_to_the_double_dots_as_the_fmaf_chain():
    """The split-bf16 pass (mhlik_batched_x3.hpp) against the f32-MFMA pass: both take the same float matrix and
    float Theta; the yardstick is the double evaluation of those float inputs.  The split pass may not be further
    from it than the float chain's own rounding allows (a few float ulps of the per-row magnitude, averaged over
    the rows), and the two passes agree with each other to that level."""
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d, K = 50000, 96, 64
    Xf = np.asfortranarray(synthetic_host(777, N, d).astype(np.float32))
    Thf = np.random.default_rng(9).normal(0, 0.2, (K, d)).astype(np.float32)
    dots = Xf.astype(np.float64) @ Thf.astype(np.float64).T
    want = -np.logaddexp(0.0, dots).sum(axis=0)
    with Likelihood(N, d, precision=2) as lk:
        lk.upload(Xf)
        lk.set_batched_f32_mode("exact")
        exact = lk.loglik_batched(Thf, dtype=np.float32)
        lk.set_batched_f32_mode("split")
        split = lk.loglik_batched(Thf, dtype=np.float32)
        again = lk.loglik_batched(Thf, dtype=np.float32)
    assert np.array_equal(split, again)  # fixed summation order: reproducible run to run
    err_exact = np.abs(exact - want) / np.abs(want)
    err_split = np.abs(split - want) / np.abs(want)
    assert err_exact.max() <= 1e-6 and err_split.max() <= 1e-6, (err_exact.max(), err_split.max())
    assert err_split.max() <= 4 * max(err_exact.max(), 5e-8), (err_exact.max(), err_split.max())


@pytest.mark.parametrize("N,d,K", [(3000, 384, 64),   # the widest Theta whose three-piece image fits LDS (155 KB, one workgroup per CU)
                                   (2000, 97, 16), (2000, 97, 17),  # chain-block boundary, ragged last k-block
                                   (700, 31, 49), (1, 5, 64), (1024, 32, 1), (5000, 160, 33)])
def test_batched_f32_split_edge_shapes_agree_with_exact_mode(N, d, K):
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(99, N, d)
    Thf = np.random.default_rng(N + d + K).normal(0, 0.5 / np.sqrt(d), (K, d)).astype(np.float32)
    want = _oracle_batched(X, Thf.astype(np.float64))
    with Likelihood(N, d, precision=2) as lk:
        lk.upload(np.asfortranarray(X.astype(np.float32)))
        lk.set_batched_f32_mode("exact")
        exact = lk.loglik_batched(Thf, dtype=np.float32)
        lk.set_batched_f32_mode("split")
        split = lk.loglik_batched(Thf, dtype=np.float32)
    assert np.all(np.abs(split - want) <= 1e-4 * np.abs(want))
    assert np.all(np.abs(split - exact) <= 2e-6 * np.abs(exact)), np.max(np.abs(split - exact) / np.abs(exact))


def test_batched_f32_split_on_an_empty_matrix():
    from mcmc_amd import Likelihood

    with Likelihood(0, 4, precision=2) as lk:
        lk.upload(np.zeros((0, 4), dtype=np.float32, order="F"))
        lk.set_batched_f32_mode("split")
        assert list(lk.loglik_batched(np.ones((3, 4), dtype=np.float32), dtype=np.float32)) == [0.0, 0.0, 0.0]


def test_batched_f32_split_falls_back_when_theta_is_too_wide_for_lds():
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d, K = 2048, 500, 64  # three-piece image of Theta: 16 k-blocks x 48 KB > 160 KB
    X = synthetic_host(5, N, d)
    Thf = np.random.default_rng(1).normal(0, 0.05, (K, d)).astype(np.float32)
    with Likelihood(N, d, precision=2) as lk:
        lk.upload(np.asfortranarray(X.astype(np.float32)))
        lk.set_batched_f32_mode("split")
        a = lk.loglik_batched(Thf, dtype=np.float32)
        lk.set_batched_f32_mode("exact")
        b = lk.loglik_batched(Thf, dtype=np.float32)
        with pytest.raises(Exception):
            lk.set_batched_f32_mode_raw(7)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_batched_begin_end_equals_the_blocking_call(dtype):
    """mhlik_loglik_batched_begin_* + _end: the same bits as the blocking call, and the context refuses any other
    evaluation in between."""
    from mcmc_amd import Likelihood
    from mcmc_amd._lib import MhlikError
    from mcmc_amd.likelihood import synthetic_host

    N, d, K = 30000, 24, 40
    X = synthetic_host(3, N, d)
    Theta = np.random.default_rng(4).normal(0, 0.2, (K, d)).astype(dtype)
    with Likelihood(N, d, precision=1 if dtype == np.float64 else 2) as lk:
        lk.upload(X if dtype == np.float64 else np.asfortranarray(X.astype(np.float32)))
        want = lk.loglik_batched(Theta, dtype=dtype)
        with pytest.raises(MhlikError):
            lk.loglik_batched_end()  # nothing begun
        lk.loglik_batched_begin(Theta, dtype=dtype)
        Theta_copy = Theta.copy()
        Theta[:] = 0  # the caller's Theta has been copied by _begin
        with pytest.raises(MhlikError):
            lk.loglik_batched(Theta_copy, dtype=dtype)
        with pytest.raises(MhlikError):
            lk.loglik_batched_begin(Theta_copy, dtype=dtype)
        with pytest.raises(MhlikError):
            (lk.loglik if dtype == np.float64 else lk.loglik_f32)(Theta_copy[0])
        with pytest.raises(MhlikError):  # nor may the matrix change under the running pass
            lk.upload(X if dtype == np.float64 else np.asfortranarray(X.astype(np.float32)))
        with pytest.raises(MhlikError):
            lk.fill_synthetic(1)
        got = lk.loglik_batched_end()
        assert np.array_equal(got, want)
        assert np.array_equal(lk.loglik_batched(Theta_copy, dtype=dtype), want)  # and the context is usable again


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_batched_overflow_column(mode):
    """Dots at and far beyond the ends of exp's range, in both epilogues of the double pass: -inf above 709.78 like
    the reference's formula, exactly 0 below the underflow point -- also for |x| = 1e30, where a range reduction
    x - n ln2 has lost every bit of r."""
    from mcmc_amd import Likelihood

    N, d = 2000, 2
    X = np.zeros((N, d), order="F")
    X[:, 0] = 1.0
    X[:, 1] = np.linspace(-1, 1, N)
    Theta = np.array([[800.0, 0.0], [0.1, 0.2], [-800.0, 0.0], [1e30, 0.0], [-1e30, 0.0], [-3e300, 0.0]])
    with Likelihood(N, d) as lk:
        lk.upload(X)
        lk.set_batched_f64_mode(mode)
        got = lk.loglik_batched(Theta)
    want = _oracle_batched(X, Theta)
    assert got[0] == want[0] == -np.inf
    assert abs(got[1] - want[1]) <= 1e-10 * abs(want[1])
    assert got[2] == want[2] == 0.0
    assert got[3] == want[3] == -np.inf
    assert got[4] == want[4] == 0.0
    assert got[5] == want[5] == 0.0


def test_multichain_sampler_reproduces_single_chains():
    """K lock-step chains over the batched kernel: chain k equals the oracle's
    cpu_sampler run with that chain's seed."""
    from mcmc_amd import Likelihood, samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d, K = 20000, 8, 5
    X = synthetic_host(12345, N, d)
    with Likelihood(N, d) as lk:
        lk.upload(X)
        out = S.multichain_sampler(lk, seeds=list(range(10, 10 + K)), Ns=120, burnin=40, rwsd=0.03)
    for k in range(K):
        ref = O.cpu_sampler(X, 10 + k, 120, 40, 0.03)
        assert np.array_equal(out["samples"][k], ref["samples"]), k
        assert np.array_equal(out["burn"][k], ref["burn"]), k
        assert out["accepted"][k] == int(ref["trace"]["accept"][39:].sum())


def test_fast_softplus_on_device():
    """mhlik_neg_softplus as compiled for gfx950 (the K-chain double kernel's
    epilogue) against a long double evaluation: tools/check_softplus."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_softplus")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stdout + out.stderr
