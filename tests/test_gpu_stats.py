"""Chain autocorrelation on the device against the
CPU restatement of src/util/processing_util.c:206-260 (oracle/ess.c), and
tune_ess_gpu (which uses it) against the oracle's tune_ess_cpu."""
import ctypes as C

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


def _ar1(n, d, rho, seed):
    rng = np.random.default_rng(seed)
    x = np.zeros((n, d))
    e = rng.normal(size=(n, d))
    for s in range(1, n):
        x[s] = rho * x[s - 1] + e[s]
    return x + rng.normal(size=d) * 3.0  # non-zero means


@pytest.mark.parametrize("n,d,lags", [(5000, 16, 500), (777, 3, 50), (1200, 64, 500), (300, 300, 40), (600, 1, 600),
                                      (1000, 5, 1)])
def test_autocorrelation_matches_oracle(n, d, lags):
    from mcmc_amd import Likelihood

    x = _ar1(n, d, 0.9, n + d)
    want = np.zeros(lags)
    wsum = O.lib().circular_autocorrelation(O.ptr(want), O.ptr(x), d, n, lags)
    with Likelihood(8, d) as lk:  # no design matrix needed
        got, gsum = lk.autocorrelation(x, lags)
    scale = np.maximum(np.abs(want), 1e-300)
    assert np.max(np.abs(got - want) / scale) <= 1e-9, np.max(np.abs(got - want) / scale)
    assert abs(gsum - wsum) <= 1e-9 * max(1.0, abs(wsum))
    ess_want = n / (1 + 2 * wsum)
    assert abs(n / (1 + 2 * gsum) - ess_want) <= 1e-8 * abs(ess_want)


def test_autocorrelation_float_twin():
    from mcmc_amd import Likelihood

    n, d, lags = 3000, 8, 200
    x = _ar1(n, d, 0.8, 5)
    want = np.zeros(lags)
    O.lib().circular_autocorrelation(O.ptr(want), O.ptr(x), d, n, lags)
    with Likelihood(8, d) as lk:
        got, _ = lk.autocorrelation(x.astype(np.float32), lags, dtype=np.float32)
    # the product of d ratios in float: compare where it is not vanishingly small
    big = np.abs(want) > 1e-6
    assert np.max(np.abs(got[big] - want[big]) / np.abs(want[big])) <= 2e-3


def test_autocorrelation_rejects_bad_arguments():
    from mcmc_amd import Likelihood
    from mcmc_amd._lib import MhlikError

    with Likelihood(8, 2) as lk:
        with pytest.raises(MhlikError):
            lk.autocorrelation(np.zeros((10, 2)), 11)  # more lags than samples


def test_tune_ess_gpu_matches_oracle():
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d, length, burnin = 400, 3, 600, 50
    X = synthetic_host(12345, N, d)
    # oracle
    L = O.lib()
    init = np.zeros(d)
    mct = O.mcmc_tune_str(rwsd=0.5, rwsdf=0.5)
    rng = O.Rng(7, L)
    mvout = np.empty(N)
    L.oracle_set_verbose(0)
    L.tune_ess_cpu(O.data_str(data=O.ptr(X).value, mvout=O.ptr(mvout).value), rng.r, O.mcmc_str(ddata=d, Nd=N, Ns=1, burnin=burnin, impl=0),
                   C.byref(mct), O.ptr(init), length)
    S.load_samplers().mcmc_amd_set_verbose(0)
    got = S.tune_ess_gpu(X, 7, burnin, 0.5, length)
    assert got == mct.rwsd, (got, mct.rwsd)
