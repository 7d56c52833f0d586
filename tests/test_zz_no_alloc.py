"""mhlik_loglik_f64_noalloc (include/mhlik.h): the double-precision likelihood
with its tile partials in the library's static device array.  No counterpart
in the reference; it is for callers that cannot allocate.  Checked through the
C-ABI against mhlik_loglik_f64 -- the same kernels, so the bar is bit equality
-- and against the CPU oracle within 1e-10, on two calls in a row, the second
on a smaller input (the array keeps the first call's contents)."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu

REL_TOL_DP = 1e-10


def test_noalloc_matches_allocating_entry_point_twice():
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    rng = np.random.default_rng(7)
    # first call: 3 000 017 rows (5860 tiles, ragged); second: 40 001 rows (79 tiles) -- the scratch still holds
    # the partials of the first call beyond tile 79
    for N, d in [(3_000_017, 24), (40_001, 24)]:
        X = synthetic_host(2024, N, d)
        th = rng.normal(0, 0.2, d)
        want, _ = O.log_likelihood(X, th)
        with Likelihood(N, d) as lk:
            lk.upload(X)
            got = lk.loglik_noalloc(th)
            ref = lk.loglik(th)
            again = lk.loglik_noalloc(th)
        assert got == ref, (got, ref)
        assert again == ref
        assert abs(got - want) <= REL_TOL_DP * abs(want), (got, want)


def test_noalloc_interleaved_contexts_and_limits():
    """Two live contexts share the device's array: calls that do not overlap
    stay correct.  Sizes beyond the stated maxima are refused with a message."""
    from mcmc_amd import Likelihood
    from mcmc_amd._lib import MhlikError
    from mcmc_amd.likelihood import synthetic_host

    Xa, Xb = synthetic_host(1, 100_003, 8), synthetic_host(2, 7_001, 8)
    th = np.linspace(-0.3, 0.3, 8)
    with Likelihood(100_003, 8) as a, Likelihood(7_001, 8) as b:
        a.upload(Xa)
        b.upload(Xb)
        ra, rb = a.loglik(th), b.loglik(th)
        for _ in range(3):
            assert a.loglik_noalloc(th) == ra
            assert b.loglik_noalloc(th) == rb
    # theta wider than the kernarg segment would need the staging buffer
    with Likelihood(600, 300) as w:
        w.upload(synthetic_host(3, 600, 300))
        with pytest.raises(MhlikError, match="n_cols=300 exceeds 256"):
            w.loglik_noalloc(np.zeros(300))
    # more row tiles than the static array has slots: d = 1 keeps the matrix small (1.07 GB)
    n_big = (1 << 18) * 512 + 1
    with Likelihood(n_big, 1) as big:
        big.fill_synthetic(5)
        with pytest.raises(MhlikError, match="exceed the static scratch"):
            big.loglik_noalloc(np.zeros(1))
        assert big.loglik(np.zeros(1)) == pytest.approx(-n_big * np.log(2.0), rel=1e-12)
