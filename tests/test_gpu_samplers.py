"""The C samplers on the MI355X vs the CPU oracle: with seeded proposals the
DP sampler must reproduce the oracle's accept/reject sequence and samples
(BASELINE.json north_star; compare_cpu_gpu-shaped, but re-seeded before each
run -- tests/compare_cpu_gpu.c:96-105 continues one RNG stream instead)."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu

REL_TOL_DP = 1e-10
REL_TOL_SP = 1e-4


@pytest.mark.parametrize("N,d,seed", [(10000, 3, 0), (10000, 3, 1), (10000, 16, 2), (50000, 16, 0), (3000, 64, 1)])
def test_gpu_sampler_trace_equals_cpu_sampler(N, d, seed):
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(12345, N, d)
    Ns, burnin, rwsd = 300, 100, 2.38 / np.sqrt(d) * 0.05
    ref = O.cpu_sampler(X, seed, Ns, burnin, rwsd)
    got = S.gpu_sampler(X, seed, Ns, burnin, rwsd)
    assert len(ref["trace"]) == len(got["trace"]) == Ns + burnin - 1
    assert np.array_equal(ref["trace"]["accept"], got["trace"]["accept"]), "accept/reject sequence differs"
    assert np.array_equal(ref["samples"], got["samples"])
    assert np.array_equal(ref["burn"], got["burn"])
    rel = np.abs(ref["trace"]["lik"] - got["trace"]["lik"]) / np.abs(ref["trace"]["lik"])
    assert rel.max() <= REL_TOL_DP
    assert ref["acceptance"] == got["acceptance"]
    # the decision margin says how close any iteration came to flipping
    margin = np.abs(ref["trace"]["logu"] - ref["trace"]["ratio"]).min()
    assert margin > 1e-9


@pytest.mark.parametrize("Ns,burnin", [(1, 1), (1, 2), (2, 1), (5000, 1)])
def test_gpu_sampler_shortest_and_chunked_chains(Ns, burnin):
    """Chain lengths at the edges: one sample after a burn-in that is only the initial evaluation
    (burn loop i = 1 .. burnin-1 is empty, mcmc_gpu.cu:563-600), and a chain longer than one device batch
    (4096 iterations per upload)."""
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(12345, 2000, 3)
    rwsd = 2.38 / np.sqrt(3) * 0.2
    ref = O.cpu_sampler(X, 3, Ns, burnin, rwsd)
    got = S.gpu_sampler(X, 3, Ns, burnin, rwsd)
    assert len(ref["trace"]) == len(got["trace"]) == Ns + burnin - 1
    assert np.array_equal(ref["trace"]["accept"], got["trace"]["accept"])
    assert np.array_equal(ref["samples"], got["samples"])
    assert np.array_equal(ref["burn"], got["burn"])
    assert ref["acceptance"] == got["acceptance"]


def test_tuner_matches_oracle():
    import ctypes as C

    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 5000, 3
    X = synthetic_host(12345, N, d)
    L = O.lib()
    mvout = np.empty(N)
    data = O.data_str(data=O.ptr(X).value, mvout=O.ptr(mvout).value)
    mcin = O.mcmc_str(ddata=d, Nd=N, Ns=1, burnin=50, impl=0)
    mct = O.mcmc_tune_str(rwsd=2.38 / np.sqrt(d), rwsdf=0)
    init = np.zeros(d)
    rng = O.Rng(7, L)
    L.tune_target_a_cpu_v2(data, rng.r, mcin, C.byref(mct), O.ptr(init), 0.25, 4)
    got = S.tune_target_a_gpu_v2(X, 7, 50, 2.38 / np.sqrt(d), 0.25, 4)
    assert got == mct.rwsd


def test_attached_context_and_device_data():
    from mcmc_amd import Likelihood, samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 20000, 8
    X = synthetic_host(4242, N, d)
    ref = O.cpu_sampler(X, 3, 100, 20, 0.02)
    with Likelihood(N, d, timing=True) as lk:
        lk.fill_synthetic(4242)
        with S.attached(lk):
            got = S.gpu_sampler(None, 3, 100, 20, 0.02, n_rows=N, n_cols=d)
    assert np.array_equal(ref["trace"]["accept"], got["trace"]["accept"])
    assert np.array_equal(ref["samples"], got["samples"])
    assert got["res"].kernelTime > 0


def test_sp_likelihood_within_1e4_of_dp_and_reference_order():
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    for N, d in [(10000, 3), (200000, 16), (65536, 64)]:
        X = synthetic_host(12345, N, d)
        Xf = np.asfortranarray(X.astype(np.float32))
        rng = np.random.default_rng(1)
        with Likelihood(N, d, precision=2) as lk:
            lk.upload(Xf)
            for _ in range(3):
                th = rng.normal(0, 0.3, d)
                dp, _ = O.log_likelihood(X, th)
                got = lk.loglik_f32(th.astype(np.float32))
                assert abs(got - dp) <= REL_TOL_SP * abs(dp)
                L = O.lib()
                scratch = np.empty(N, dtype=np.float32)
                thf = th.astype(np.float32)
                ref_sp = L.oracle_gpu_likelihood_f(O.ptr(Xf), N, d, N, O.ptr(thf), O.ptr(scratch), 256, 32)
                assert abs(got - ref_sp) <= REL_TOL_SP * abs(ref_sp)


def test_sp_sampler_runs_and_agrees_statistically():
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 20000, 3
    X = synthetic_host(12345, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    dp = S.gpu_sampler(X, 5, 2000, 500, 0.05, trace=False)
    sp = S.sp_sampler(Xf, 5, 2000, 500, 0.05, trace=False)
    assert 0.02 < sp["acceptance"] < 0.98
    m_dp, m_sp = dp["samples"].mean(0), sp["samples"].mean(0)
    s_dp = dp["samples"].std(0)
    assert np.all(np.abs(m_dp - m_sp) < 3 * s_dp + 0.05)


def test_config2_shape_compare_cpu_gpu_trace():
    """BASELINE configs[1]: N=1M, d=16 DP, accept/reject parity (tests/compare_cpu_gpu.c shape,
    with the RNG re-seeded before each run)."""
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 1_000_000, 16
    X = synthetic_host(12345, N, d)
    Ns, burnin, rwsd = 150, 50, 2.38 / np.sqrt(d) * 5e-3
    ref = O.cpu_sampler(X, 0, Ns, burnin, rwsd)
    got = S.gpu_sampler(X, 0, Ns, burnin, rwsd)
    assert np.array_equal(ref["trace"]["accept"], got["trace"]["accept"])
    assert np.array_equal(ref["samples"], got["samples"])
    rel = np.abs(ref["trace"]["lik"] - got["trace"]["lik"]) / np.abs(ref["trace"]["lik"])
    assert rel.max() <= REL_TOL_DP
    assert 0.05 < got["acceptance"] < 0.95


def test_config2_contested_decisions_trace_equals_cpu_sampler():
    """BASELINE configs[1] where parity can fail: N=1M, d=16, the chain is first walked on the GPU and its step size
    rescaled until the median |log ratio| of a run is a few nats (log u and the ratio are then of the same size -- a
    chain started at theta = 0 with small steps decides everything by hundreds of nats, and so does a chain that is
    still climbing with large steps), then the same start and step size go to the oracle's
    cpu_sampler and to gpu_sampler (tests/compare_cpu_gpu.c:89-110 shape, re-seeded; accept test mcmc_cpu.c:57-60)."""
    from mcmc_amd import Likelihood, samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 1_000_000, 16
    X = synthetic_host(12345, N, d)
    with Likelihood(N, d) as lk:
        lk.upload(X)
        with S.attached(lk):
            sd0 = 2.38 / np.sqrt(d) * 5e-3
            warm = S.gpu_sampler(None, 11, 2000, 1, sd0, n_rows=N, n_cols=d, trace=False)  # 2000 iterations away from 0
            theta, sd, acc, walk_med = S.walk_to_contested(N, d, 0, warm["samples"][-1], sd0)
            assert 0.15 <= acc <= 0.6 and 0.5 <= walk_med <= 10.0, (acc, walk_med, sd)
            got = S.gpu_sampler(None, 5, 120, 1, sd, theta0=theta, n_rows=N, n_cols=d)
    ref = O.cpu_sampler(X, 5, 120, 1, sd, theta0=theta)
    ct, gt = ref["trace"], got["trace"]
    assert len(ct) == len(gt) == 120
    med = float(np.median(np.abs(ct["ratio"])))
    margin = np.abs(ct["logu"] - ct["ratio"])
    worst = int(np.argmin(margin))
    rel = np.abs(ct["lik"] - gt["lik"]) / np.abs(ct["lik"])
    print("contested: sd=%.3g acceptance=%.3f median|ratio|=%.3g min margin=%.3g (|dL| there %.3g) max rel dL=%.3g"
          % (sd, ct["accept"].mean(), med, margin.min(), abs(ct["lik"][worst] - gt["lik"][worst]), rel.max()))
    assert 0.05 <= med <= 30.0, "decisions are not contested: median |ratio| = %g" % med
    assert 0.1 <= ct["accept"].mean() <= 0.7
    assert np.array_equal(ct["accept"], gt["accept"]), "accept/reject sequence differs"
    assert np.array_equal(ref["samples"], got["samples"])
    assert rel.max() <= REL_TOL_DP
    # the closest call was still decided by far more than the two sums differ
    assert margin.min() > 100 * np.abs(ct["lik"] - gt["lik"]).max()


def test_config1_shape_repeated_runs_share_one_stream():
    """BASELINE configs[0] (tests/multiple_runs_cpu.c:56-71): back-to-back sampler runs on one
    data set with the RNG state carried across runs; GPU and CPU agree run for run."""
    from mcmc_amd import Likelihood, samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 10000, 3
    X = synthetic_host(12345, N, d)
    rng_cpu, rng_gpu = O.Rng(0), S.Rng(0)
    import ctypes as C

    L = O.lib()
    with Likelihood(N, d) as lk:
        lk.upload(X)
        with S.attached(lk):
            for run in range(4):
                g = S.gpu_sampler(X, None, 120, 30, 0.08, rng=rng_gpu)
                mvout = np.empty(N)
                samples, burn = np.zeros((120, d)), np.zeros((30, d))
                res = O.out_str()
                L.cpu_sampler(O.data_str(data=O.ptr(X).value, mvout=O.ptr(mvout).value), rng_cpu.r,
                              O.mcmc_str(ddata=d, Nd=N, Ns=120, burnin=30), O.mcmc_tune_str(rwsd=0.08),
                              O.mcmc_v_str(samples=O.ptr(samples).value, burn=O.ptr(burn).value), C.byref(res))
                assert np.array_equal(samples, g["samples"]), run
                assert res.acceptance == g["acceptance"]


def test_cpu_path_entry_points_run_on_the_gpu_and_match_the_oracle():
    """cpu_sampler / log_likelihood of the product library (the reference's
    mcmc_cpu.h names) against the oracle's CPU restatement of the same names."""
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 30_000, 5
    X = synthetic_host(12345, N, d)
    ref = O.cpu_sampler(X, 4, 120, 30, 0.03)
    run = S.cpu_sampler(X, 4, 120, 30, 0.03)
    assert np.array_equal(ref["trace"]["accept"], run["trace"]["accept"])
    assert np.array_equal(ref["samples"], run["samples"])
    assert np.array_equal(ref["burn"], run["burn"])
    rng = np.random.default_rng(0)
    for X2 in (X, synthetic_host(7, 5000, 3)):  # a second matrix: the kept buffers must be replaced
        for _ in range(2):
            th = rng.normal(0, 0.2, X2.shape[1])
            want, _ = O.log_likelihood(X2, th)
            got = S.log_likelihood(X2, th)
            assert abs(got - want) <= 1e-10 * abs(want)


def test_log_likelihood_sees_a_matrix_refilled_in_place():
    """The reference's log_likelihood reads data.data on every call (mcmc_cpu.c:419-445), so a caller may refill the
    matrix in place, or free it and get the same address back (the Nd sweeps of tests/large_data_cpu.c): the product
    must evaluate what is in host memory NOW, never a copy kept from an earlier call."""
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 40_000, 6
    X = synthetic_host(21, N, d)
    th = np.random.default_rng(2).normal(0, 0.3, d)
    first = S.log_likelihood(X, th)
    assert abs(first - O.log_likelihood(X, th)[0]) <= 1e-10 * abs(first)
    X[:] = synthetic_host(22, N, d)  # same pointer, same shape, new contents
    want, _ = O.log_likelihood(X, th)
    got = S.log_likelihood(X, th)
    assert got != first
    assert abs(got - want) <= 1e-10 * abs(want)
    X[::7, 1] *= -1.0  # a sparse change no strided sample would be sure to see
    want, _ = O.log_likelihood(X, th)
    got = S.log_likelihood(X, th)
    assert abs(got - want) <= 1e-10 * abs(want)


@pytest.mark.parametrize("host_loop", [False, True])
@pytest.mark.parametrize("N,d,scale", [(10000, 3, 0.3), (40000, 16, 0.05), (20000, 64, 0.02)])
def test_sp_sampler_host_arithmetic_against_teacher_forced_oracle(monkeypatch, N, d, scale, host_loop):
    """sp_sampler's arithmetic AROUND the likelihood (mcmc_gpu_sp.cu:86-177, 561-580: float state and proposals, float
    prior accumulator over double log(pdf) terms, float posterior and ratio, logf(uf) <= ratio, rows written) against
    an independent restatement, oracle/sp_path.c.  In float the N-term sum legitimately depends on the order, so the
    oracle is fed the product's float likelihoods ("teacher forcing") and each fed value is checked against the DP
    oracle at the ORACLE's own proposal (1e-4): a product whose state, stream or proposals differed would feed
    likelihoods of other points."""
    from mcmc_amd import Likelihood
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    if host_loop:
        monkeypatch.setenv("MCMC_AMD_HOST_LOOP", "1")
    X = synthetic_host(4242, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    Ns, burnin, seed = 120, 40, 9
    rwsdf = 2.38 / np.sqrt(d) * scale
    theta0 = np.random.default_rng(1).normal(0, 0.05, d).astype(np.float32)
    with Likelihood(N, d, precision=2) as lf:
        lf.upload(Xf)
        lik0 = lf.loglik_f32(theta0)
        with S.attached(lf):
            got = S.sp_sampler(None, seed, Ns, burnin, rwsdf, theta0=theta0, n_rows=N, n_cols=d)
    gt = got["trace"]
    assert len(gt) == burnin - 1 + Ns
    feed = np.concatenate([[lik0], gt["lik"]]).astype(np.float32)
    assert np.array_equal(feed[1:].astype(np.float64), gt["lik"])  # the trace holds float values
    ref = O.sp_sampler_forced(seed, d, rwsdf, burnin, Ns, feed, theta0=theta0)
    rt = ref["trace"]
    assert np.array_equal(rt["logu"], gt["logu"])
    assert np.array_equal(rt["accept"], gt["accept"]) and 0 < rt["accept"].sum() < len(rt)
    if host_loop:
        assert np.array_equal(rt["ratio"], gt["ratio"])
    else:  # the device's double log(pdf) may differ from glibc's in the last place: at most one float ulp of the ratio
        assert np.allclose(rt["ratio"], gt["ratio"], rtol=2.4e-7, atol=0)
    assert np.array_equal(ref["samples"], got["samples"])
    assert np.array_equal(ref["burn"], got["burn"])
    assert ref["accepted"] == int(round(got["acceptance"] * Ns))
    # every fed likelihood is the likelihood of the oracle's own proposal
    for i in range(len(rt)):
        dp, _ = O.log_likelihood(X, ref["proposals"][i].astype(np.float64))
        assert abs(float(feed[i + 1]) - dp) <= REL_TOL_SP * abs(dp), i
    dp0, _ = O.log_likelihood(X, theta0.astype(np.float64))
    assert abs(lik0 - dp0) <= REL_TOL_SP * abs(dp0)
