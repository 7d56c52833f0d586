"""DP likelihood on the MI355X vs the CPU oracle (restatement of
src/mcmc_cpu/mcmc_cpu.c:419-445).  Bar: row dots bit-exact, L within 1e-10
relative (BASELINE.json north_star)."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu

REL_TOL_DP = 1e-10  # north_star: "log-likelihood agreement within 1e-10 relative in DP"


def _theta(d, seed=0, scale=0.3):
    rng = np.random.default_rng(seed)
    return rng.normal(0, scale, d)


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("N,d", [(10000, 3), (4097, 16), (1024, 64), (100003, 16), (513, 1), (1, 5), (30000, 128)])
def test_loglik_matches_oracle(N, d, layout):
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(12345, N, d)
    with Likelihood(N, d, layout=layout) as lk:
        lk.upload(X)
        for s in range(3):
            th = _theta(d, s)
            want, mvout = O.log_likelihood(X, th)
            got = lk.loglik(th)
            assert abs(got - want) <= REL_TOL_DP * abs(want), (got, want)
            dots = lk.rowdots(th)
            assert np.array_equal(dots, mvout), "row dots must be bit-identical to the gslcblas order"


def test_device_fill_equals_host_generator():
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d = 70001, 24
    X = synthetic_host(777, N, d)
    with Likelihood(N, d, precision=3) as lk:
        lk.fill_synthetic(777)
        assert np.array_equal(lk.download(np.float64), X)
        assert np.array_equal(lk.download(np.float32), X.astype(np.float32))


@pytest.mark.parametrize("variant", list(range(1, 13)))
def test_every_variant(variant):
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d = 20011, 37
    X = synthetic_host(5, N, d)
    th = _theta(d, 3)
    want, mvout = O.log_likelihood(X, th)
    for layout in (1, 2):
        with Likelihood(N, d, layout=layout, variant=variant) as lk:
            lk.upload(X)
            got = lk.loglik(th)
            assert abs(got - want) <= REL_TOL_DP * abs(want)


def test_overflow_and_extremes_like_reference():
    """-log(1+exp(x)) is naive in the reference (mcmc_cpu.c:440): x > 709.78
    gives -inf, very negative x gives -0."""
    from mcmc_amd import Likelihood

    N, d = 1000, 2
    X = np.zeros((N, d), order="F")
    X[:, 0] = 1.0
    X[:, 1] = np.linspace(-1, 1, N)
    with Likelihood(N, d) as lk:
        lk.upload(X)
        for th in ([800.0, 0.0], [-800.0, 0.0], [0.0, 0.0], [700.0, 5.0]):
            want, _ = O.log_likelihood(X, np.array(th))
            got = lk.loglik(np.array(th))
            if np.isinf(want):
                assert got == want
            else:
                assert abs(got - want) <= REL_TOL_DP * abs(want) + 1e-300


def test_deterministic_and_segment_invariant():
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d = 50000, 8
    X = synthetic_host(9, N, d)
    th = _theta(d, 1)
    with Likelihood(N, d) as lk:
        lk.upload(X)
        a = lk.loglik(th)
        b = lk.loglik(th)
        assert a == b
        segs = lk.segment_sums()
        tot = 0.0
        for s in segs:
            tot += s
        assert tot == a


def test_many_tiles_uses_wide_segment_sum():
    """More than 8192 tiles per segment switches the segment reduction to one
    workgroup per segment (N ~ 1e8 shapes); same tolerance, still deterministic."""
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d = 8 * 8193 * 512 + 77, 2
    X = synthetic_host(3, N, d)
    th = np.array([0.3, -0.2])
    want, _ = O.log_likelihood(X, th)
    with Likelihood(N, d) as lk:
        lk.upload(X)
        a = lk.loglik(th)
        b = lk.loglik(th)
    assert a == b
    assert abs(a - want) <= REL_TOL_DP * abs(want)


def test_empty_matrix_and_wide_theta():
    """N = 0: the sum over no rows is 0 (mcmc_cpu.c:438-441 starts from 0);
    d > 256: theta no longer fits the kernarg segment and goes through the
    pinned staging buffer."""
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    with Likelihood(0, 4, precision=3) as lk:
        lk.upload(np.zeros((0, 4), order="F"))
        lk.upload(np.zeros((0, 4), order="F", dtype=np.float32))
        assert lk.loglik(np.ones(4)) == 0.0
        assert lk.loglik_f32(np.ones(4, dtype=np.float32)) == 0.0
        assert list(lk.loglik_batched(np.ones((3, 4)))) == [0.0, 0.0, 0.0]
    N, d = 3000, 300
    X = synthetic_host(11, N, d)
    th = np.random.default_rng(4).normal(0, 0.05, d)
    want, mvout = O.log_likelihood(X, th)
    with Likelihood(N, d) as lk:
        lk.upload(X)
        got = lk.loglik(th)
        dots = lk.rowdots(th)
    assert np.array_equal(dots, mvout)
    assert abs(got - want) <= REL_TOL_DP * abs(want)
