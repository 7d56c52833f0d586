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


def _mp_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        block_rows = int(rng.choice([64, 128, 256, 512, 1024]))
        N = int(rng.integers(block_rows, 40 * block_rows if block_rows <= 256 else 12 * block_rows))
        if i % 3 == 0:
            N = (N // block_rows) * block_rows  # no ragged last block
        d = int(rng.choice([1, 3, 8, 16, 17, 40, 64]))
        F = int(rng.choice([2, 3, 10, 16]))
        K = int(rng.choice([1, 5, 16, 33, 64]))
        shared = bool(i % 2)
        out.append((N, d, block_rows, F, K, shared, int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("N,d,block_rows,F,K,shared,seed", _mp_cases(14, 77))
def test_random_mixed_precision_sequences(N, d, block_rows, F, K, shared, seed):
    """Mixed precision (mcmc_mp.cu:28-200,475-552) on random geometries: a sequence of calls -- the indicators evolve
    from call to call -- of the single-chain path against the oracle (identical bright / dark counts, value within
    2e-6), and of the K-chain path against K single-chain contexts' worth of oracle sequences (chain k = the oracle's
    sequence with chain k's theta, phase and seed; both float modes)."""
    import ctypes as C

    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host
    from test_gpu_mp import _oracle_mp

    rng = np.random.default_rng(seed)
    X = synthetic_host(seed, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    blocks = (N + block_rows - 1) // block_rows
    eps, calls = 1.7e-7, 5
    thetas = rng.normal(0, 0.4 / np.sqrt(d), (calls, K, d))
    phases = rng.integers(0, F, (calls, K))
    if shared:
        phases[:] = phases[:, :1]
    seeds = (1000 + np.arange(K)).astype(np.uint64)
    # oracle: one sequence per chain
    want = np.empty((calls, K))
    wb = np.empty((calls, K), dtype=np.int64)
    for k in range(K if K <= 5 else 3):  # the oracle's scalar loop is slow: all chains when few, else three of them
        z = np.ones(blocks, dtype=np.int8)
        for c in range(calls):
            want[c, k], wb[c, k], _ = _oracle_mp(X, Xf, thetas[c, k], block_rows, blocks, F, int(phases[c, k]),
                                                 int(seeds[k]), c, eps, z)
    checked = range(K if K <= 5 else 3)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.upload(X)
        lk.upload(Xf)
        lk.mp_reset(block_rows, blocks)
        for c in range(calls):  # chain 0 through the single-chain entry point
            got, gb, gd = lk.mp_loglik(thetas[c, 0], block_rows, blocks, F, int(phases[c, 0]), int(seeds[0]), c, eps)
            assert gb == wb[c, 0] and gb + gd == blocks, (c, gb, gd, wb[c, 0])
            assert abs(got - want[c, 0]) <= 2e-6 * abs(want[c, 0]) + 1e-9
        for mode in ("exact", "split"):
            lk.set_batched_f32_mode(mode)
            lk.mp_reset_batched(block_rows, blocks, K)
            for c in range(calls):
                vals, nb, nd = lk.mp_loglik_batched(thetas[c], F, phases[c], seeds, c, eps)
                for k in checked:
                    assert nb[k] == wb[c, k] and nb[k] + nd[k] == blocks, (mode, c, k, nb[k], wb[c, k])
                    tol = 2e-6 if mode == "exact" else 2e-5
                    assert abs(vals[k] - want[c, k]) <= tol * abs(want[c, k]) + 1e-9, (mode, c, k, vals[k], want[c, k])


def _chain_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        N = int(rng.choice([1, 511, 513, 2000, 7777, 30000]))
        d = int(rng.choice([1, 3, 16, 33, 64]))
        Ns = int(rng.choice([1, 17, 130, 400]))
        burnin = int(rng.choice([1, 2, 60]))
        step = float(rng.choice([0.02, 0.3, 2.0]))  # from every proposal accepted to nearly none
        out.append((N, d, Ns, burnin, step, int(rng.integers(1 << 20))))
    return out


@pytest.mark.parametrize("N,d,Ns,burnin,step,seed", _chain_cases(12, 5))
def test_random_chains_reproduce_the_oracle(N, d, Ns, burnin, step, seed):
    """gpu_sampler (mcmc_gpu.cu:20-128) and chain 0 / chain K-1 of the K-chain sampler against the oracle's cpu_sampler
    on random sizes, lengths and step sizes: accept / reject sequence, samples and burn-in rows identical."""
    from mcmc_amd import Likelihood, samplers as S
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(seed, N, d)
    rwsd = 2.38 / np.sqrt(d) * step
    ref = O.cpu_sampler(X, seed, Ns, burnin, rwsd)
    got = S.gpu_sampler(X, seed, Ns, burnin, rwsd)
    assert np.array_equal(ref["trace"]["accept"], got["trace"]["accept"])
    assert np.array_equal(ref["samples"], got["samples"])
    assert np.array_equal(ref["burn"], got["burn"])
    if len(ref["trace"]):
        lik_r, lik_g = ref["trace"]["lik"], got["trace"]["lik"]
        fin = np.isfinite(lik_r)
        assert np.array_equal(fin, np.isfinite(lik_g))
        assert np.all(np.abs(lik_r[fin] - lik_g[fin]) <= 1e-10 * np.abs(lik_r[fin]))
    K = 1 + seed % 7
    seeds = [seed + 3 * k for k in range(K)]
    with Likelihood(N, d) as lk:
        lk.upload(X)
        many = S.multichain_sampler(lk, seeds, Ns, burnin, rwsd)
    assert np.array_equal(ref["samples"], many["samples"][0])
    last = O.cpu_sampler(X, seeds[-1], Ns, burnin, rwsd)
    assert np.array_equal(last["samples"], many["samples"][K - 1])
    assert np.array_equal(last["burn"][1:], many["burn"][K - 1][1:])
