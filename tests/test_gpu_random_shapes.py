"""Randomised shapes through the C-ABI against the oracle (SURVEY.md section 8 rows a1-a5 and the K-chain passes):
the parametrised tests elsewhere pick their shapes by hand; this sweep draws N, d, K, the segment count and the scale
of theta from a seeded generator, so that ragged tiles, one-row shards, columns beyond a
load batch and chain counts off the 16-chain block all come up.  Bars: row dots bit-exact, L within 1e-10 (double) /
1e-4 of the double value (float), K-chain passes within the same bars of K single evaluations; segment counts
1 / 2 / 4 / 8 give sums that differ only by rounding (each is world-invariant by construction)."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = i % 4
        if kind == 0:  # around tile and block edges
            N = int(rng.choice([1, 2, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 2047, 4097]))
        elif kind == 1:
            N = int(rng.integers(1, 3000))
        else:
            N = int(rng.integers(3000, 60000))
        d = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 64, 65, 100, 128, 129, 200, 256]) if kind != 2 else rng.integers(1, 80))
        K = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64]))
        seg = int(rng.choice([1, 2, 4, 8]))
        scale = float(rng.choice([0.01, 0.3, 3.0]))
        out.append((N, d, K, seg, scale, int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("N,d,K,seg,scale,seed", _cases(28, 20240611))
def test_random_shape(N, d, K, seg, scale, seed):
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    rng = np.random.default_rng(seed)
    X = synthetic_host(seed, N, d)
    Xup = X
    theta = rng.normal(0, scale / np.sqrt(d), d)
    Theta = rng.normal(0, scale / np.sqrt(d), (K, d))
    Theta[0] = theta
    want, mv = O.log_likelihood(X, theta)
    wantK = np.array([O.log_likelihood(X, t)[0] for t in Theta])
    with Likelihood(N, d, precision=3, segments=seg) as lk:
        lk.upload(Xup)
        lk.upload(np.asfortranarray(X.astype(np.float32)))
        got = lk.loglik(theta)
        dots = lk.rowdots(theta)
        gotf = lk.loglik_f32(theta.astype(np.float32))
        fast = lk.loglik_batched(Theta)
        lk.set_batched_f64_mode("exact")
        exact = lk.loglik_batched(Theta)
        lk.set_batched_f32_mode("split")
        split = lk.loglik_batched(Theta.astype(np.float32), dtype=np.float32)
        lk.set_batched_f32_mode("exact")
        f32x = lk.loglik_batched(Theta.astype(np.float32), dtype=np.float32)
    assert np.array_equal(dots, mv)  # the gslcblas order, bit for bit
    tol = 1e-10 * abs(want) + 1e-300
    assert abs(got - want) <= tol, (got, want)
    assert abs(gotf - want) <= 1e-4 * abs(want) + 1e-6
    for name, v, rel in (("fast", fast, 1e-10), ("exact", exact, 1e-10), ("split", split, 1e-4), ("f32 mfma", f32x, 1e-4)):
        err = np.abs(v - wantK) / np.maximum(np.abs(wantK), 1e-300)
        assert np.all(err <= rel + (1e-6 / np.maximum(np.abs(wantK), 1e-300) if rel > 1e-9 else 0)), (name, err.max())
    assert abs(fast[0] - got) <= tol and abs(exact[0] - got) <= tol
