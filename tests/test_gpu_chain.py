"""The MH chain kept on the device (mhlik_chain_*)
against the per-iteration host loop and the CPU oracle.  The device runs the
same iterations from the same host-drawn random numbers, so proposals, samples
and the accept/reject sequence must be identical; the likelihood bits are the
same kernel's; only the prior (device exp/log vs glibc) may differ in the
last place."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


def _both_modes(monkeypatch, fn):
    monkeypatch.setenv("MCMC_AMD_HOST_LOOP", "1")
    host = fn()
    monkeypatch.delenv("MCMC_AMD_HOST_LOOP")
    dev = fn()
    return host, dev


@pytest.mark.parametrize("N,d,segments", [(10000, 3, 0), (30000, 16, 0), (4000, 64, 32), (2500, 300, 0)])
def test_device_chain_equals_host_loop(monkeypatch, N, d, segments):
    from mcmc_amd import Likelihood, samplers as S
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(777, N, d)
    Ns, burnin, rwsd = 200, 40, 2.38 / np.sqrt(d) * 0.05

    def run():
        with Likelihood(N, d, segments=segments, timing=True) as lk:
            lk.upload(X)
            with S.attached(lk):
                return S.gpu_sampler(None, 5, Ns, burnin, rwsd, n_rows=N, n_cols=d)

    host, dev = _both_modes(monkeypatch, run)
    assert np.array_equal(host["trace"]["accept"], dev["trace"]["accept"])
    assert np.array_equal(host["samples"], dev["samples"])
    assert np.array_equal(host["burn"], dev["burn"])
    assert np.array_equal(host["trace"]["lik"], dev["trace"]["lik"]), "same kernel, same theta bits -> same L bits"
    assert np.array_equal(host["trace"]["logu"], dev["trace"]["logu"])
    assert np.allclose(host["trace"]["ratio"], dev["trace"]["ratio"], rtol=0, atol=1e-9)
    assert host["acceptance"] == dev["acceptance"]
    assert dev["res"].kernelTime > 0


def test_device_chain_equals_oracle_and_tuner(monkeypatch):
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 20000, 8
    X = synthetic_host(4242, N, d)
    ref = O.cpu_sampler(X, 3, 500, 100, 0.02)
    got = S.gpu_sampler(X, 3, 500, 100, 0.02)
    assert np.array_equal(ref["trace"]["accept"], got["trace"]["accept"])
    assert np.array_equal(ref["samples"], got["samples"])
    rel = np.abs(ref["trace"]["lik"] - got["trace"]["lik"]) / np.abs(ref["trace"]["lik"])
    assert rel.max() <= 1e-10
    host, dev = _both_modes(monkeypatch, lambda: S.tune_target_a_gpu_v2(X, 7, 50, 2.38 / np.sqrt(d), 0.25, 3))
    assert host == dev


def test_chain_abi_begin_run_state():
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d, sd = 12000, 5, 10.0
    X = synthetic_host(99, N, d)
    rng = np.random.default_rng(0)
    th0 = rng.normal(0, 0.1, d)
    n = 64
    inc = rng.normal(0, 0.02, (n, d))
    logu = np.log(rng.uniform(size=n))
    with Likelihood(N, d) as lk:
        lk.upload(X)
        cp = lk.chain_begin(th0, sd)
        L0 = lk.loglik(th0)
        prior0 = sum(np.log(np.exp(-(x / sd) ** 2 / 2) / (np.sqrt(2 * np.pi) * sd)) for x in th0)
        assert abs(cp - (prior0 + L0)) <= 1e-12 * abs(cp)
        out = lk.chain_run(inc, logu)
        # replay on the host with the library's own single-theta likelihood
        cur, cpost = th0.copy(), cp
        for i in range(n):
            prop = cur + inc[i]
            Lp = lk.loglik(prop)
            assert Lp == out["lik"][i]
            pp = sum(np.log(np.exp(-(x / sd) ** 2 / 2) / (np.sqrt(2 * np.pi) * sd)) for x in prop) + Lp
            a = logu[i] <= pp - cpost
            assert bool(out["accepted"][i]) == bool(a)
            if a:
                cur, cpost = prop, pp
            assert np.array_equal(out["samples"][i], cur)
        cur_dev, cp_dev = lk.chain_state()
        assert np.array_equal(cur_dev, cur)
        assert abs(cp_dev - cpost) <= 1e-12 * abs(cpost)
        assert out["n_accepted"] == int(out["accepted"].sum())
        # a second batch continues from the device-side state; resume restarts from a host-side one
        out2 = lk.chain_run(inc[:8], logu[:8])
        lk.chain_resume(cur, cp_dev, sd)
        out3 = lk.chain_run(inc[:8], logu[:8])
        assert np.array_equal(out2["samples"], out3["samples"])


def test_chain_over_forced_rccl_is_bitwise_equal(monkeypatch):
    """One-rank communicator (MHLIK_FORCE_COMM=1): segment kernel -> ncclAllReduce
    on the stream -> step kernel, the path N ranks take."""
    from mcmc_amd import Likelihood

    N, d = 100_000, 16
    rng = np.random.default_rng(2)
    inc = rng.normal(0, 0.01, (50, d))
    logu = np.log(rng.uniform(size=50))

    def run(force):
        with Likelihood(N, d, device=0) as lk:
            lk.fill_synthetic(31)
            if force:
                lk.comm_init(Likelihood.make_comm_id())
            cp = lk.chain_begin(np.zeros(d), 10.0)
            return cp, lk.chain_run(inc, logu)

    cp0, plain = run(False)
    monkeypatch.setenv("MHLIK_FORCE_COMM", "1")
    cp1, forced = run(True)
    assert cp0 == cp1
    for k in ("samples", "lik", "ratio", "accepted"):
        assert np.array_equal(plain[k], forced[k]), k


@pytest.mark.parametrize("N,d", [(10000, 3), (40000, 16)])
def test_sp_device_chain_equals_host_loop(monkeypatch, N, d):
    """Float chain (sp_sampler): float proposals, float prior accumulator fed
    double log(pdf) terms, float posterior / ratio, logf(uf) <= ratio."""
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    Xf = np.asfortranarray(synthetic_host(555, N, d).astype(np.float32))
    host, dev = _both_modes(monkeypatch, lambda: S.sp_sampler(Xf, 11, 300, 50, 2.38 / np.sqrt(d) * 0.05))
    assert np.array_equal(host["trace"]["lik"], dev["trace"]["lik"])
    assert np.array_equal(host["trace"]["logu"], dev["trace"]["logu"])
    # the float ratio is quantised at ~1e-3 here; a last-place difference of the
    # device's double log(pdf) can move it by one float ulp at most
    assert np.allclose(host["trace"]["ratio"], dev["trace"]["ratio"], rtol=2e-7, atol=0)
    assert np.array_equal(host["trace"]["accept"], dev["trace"]["accept"])
    assert np.array_equal(host["samples"], dev["samples"])
    assert np.array_equal(host["burn"], dev["burn"])
    assert host["acceptance"] == dev["acceptance"]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_chain_submit_collect_pipeline_equals_run(dtype):
    """mhlik_chain_submit / _collect with two batches in flight (and a slot that grows while the other one is in
    flight) give the bits of one blocking mhlik_chain_run; begin / resume / run / state refuse meanwhile."""
    from mcmc_amd import Likelihood
    from mcmc_amd._lib import MhlikError

    N, d = 30000, 7
    rng = np.random.default_rng(5)
    sizes = [64, 256, 700, 3, 4096, 1]
    n = sum(sizes)
    inc = rng.normal(0, 0.01, (n, d)).astype(dtype)
    logu = np.log(rng.uniform(size=n)).astype(dtype)
    with Likelihood(N, d, precision=3) as lk:
        lk.fill_synthetic(77)
        cp = lk.chain_begin(np.zeros(d, dtype=dtype), 10.0, dtype=dtype)
        ref = lk.chain_run(inc, logu, dtype=dtype)
        ref_state = lk.chain_state(dtype=dtype)

        lk.chain_resume(np.zeros(d, dtype=dtype), cp, 10.0, dtype=dtype)
        assert lk.chain_state(dtype=dtype)[1] == cp  # resume alone: the host-side state, no device operation
        outs, fly, at = [], [], 0
        for m in sizes:
            if len(fly) == 2:
                outs.append(lk.chain_collect(fly.pop(0), dtype=dtype))
            lk.chain_submit(inc[at:at + m], logu[at:at + m], dtype=dtype)
            fly.append(m)
            at += m
        with pytest.raises(MhlikError, match="in flight"):
            lk.chain_submit(inc[:1], logu[:1], dtype=dtype)
        with pytest.raises(MhlikError, match="in flight"):
            lk.chain_state(dtype=dtype)
        with pytest.raises(MhlikError, match="in flight"):
            lk.chain_run(inc[:1], logu[:1], dtype=dtype)
        with pytest.raises(MhlikError, match="oldest batch"):
            lk.chain_collect(fly[0] + 1, dtype=dtype)
        while fly:
            outs.append(lk.chain_collect(fly.pop(0), dtype=dtype))
        with pytest.raises(MhlikError, match="no chain batch in flight"):
            lk.chain_collect(1, dtype=dtype)
        for k in ("samples", "lik", "ratio", "accepted"):
            assert np.array_equal(np.concatenate([o[k] for o in outs]), ref[k]), k
        assert sum(o["n_accepted"] for o in outs) == ref["n_accepted"]
        st = lk.chain_state(dtype=dtype)
        assert np.array_equal(st[0], ref_state[0]) and st[1] == ref_state[1]
        with pytest.raises(MhlikError, match="1..4096"):
            lk.chain_submit(np.zeros((4097, d), dtype=dtype), np.zeros(4097, dtype=dtype), dtype=dtype)
        # a batch submitted without a samples array cannot hand rows out
        lk.chain_submit(inc[:5], logu[:5], dtype=dtype, want_samples=False)
        with pytest.raises(MhlikError, match="without a samples array"):
            lk.chain_collect(5, dtype=dtype)
        o = lk.chain_collect(5, dtype=dtype, want_samples=False)
        assert o["samples"] is None and len(o["lik"]) == 5


@pytest.mark.parametrize("persist", ["0", "1"])
def test_run_after_burn_in_allocates_nothing(monkeypatch, persist):
    """The driver's shape: a 4-iteration burn-in, then 20 timed iterations.  The second call must find every buffer
    in place: free device memory does not move, and neither does it over a longer run that fits the slots."""
    import torch

    monkeypatch.setenv("MHLIK_CHAIN_PERSIST", persist)

    from mcmc_amd import Likelihood

    N, d = 20000, 16
    rng = np.random.default_rng(1)
    inc = rng.normal(0, 0.01, (256, d))
    logu = np.log(rng.uniform(size=256))
    with Likelihood(N, d, timing=True) as lk:
        lk.fill_synthetic(5)
        lk.chain_resume(np.zeros(d), -1e4, 10.0)
        lk.chain_run(inc[:4], logu[:4])
        free0 = torch.cuda.mem_get_info(0)[0]
        lk.chain_resume(np.zeros(d), -1e4, 10.0)
        a = lk.chain_run(inc[:20], logu[:20])
        # two-kernel path: every fourth launch of a short batch is bracketed with events (every eighth of a long one);
        # persistent launch: one pair around the batch, counted as its 20 passes
        assert a["kernel_samples"] == (5 if persist == "0" else 20)
        assert torch.cuda.mem_get_info(0)[0] == free0
        # a longer run that fits the slots: the library allocates nothing either; the HIP runtime may grow its own
        # kernel-argument pool once (512 launches with 2 KB of kernarg each; seen: one 2 MiB chunk)
        lk.chain_run(inc, logu)
        assert free0 - torch.cuda.mem_get_info(0)[0] <= 8 << 20


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N,d,segments", [(10000, 3, 8), (1000, 1, 8), (40000, 16, 8), (300_000, 7, 8), (70_001, 33, 16),
                                          (513, 5, 1), (3_000_000, 4, 8), (1_000_000, 16, 8), (2_000_003, 9, 4)])
def test_persistent_chain_kernel_equals_two_kernel_path(monkeypatch, N, d, segments, dtype):
    """chain_persist_kernel (the whole batch in one launch: every workgroup gathers every job's lane sum and decides
    for itself) against the two-kernel-per-iteration path: every output bit for bit, over several batches, ragged tiles,
    lanes with one tile and lanes with many, one segment and sixteen; tiles held in registers (d <= 16), in LDS and
    re-read from memory (3M x 4: 8 + 8 + 7 per workgroup; 1M x 16: 5 + 2 + 1), and d = 33 with nothing resident."""
    from mcmc_amd import Likelihood

    rng = np.random.default_rng(N + d)
    n = 150
    inc = rng.normal(0, 0.02, (n, d)).astype(dtype)
    logu = np.log(rng.uniform(size=n)).astype(dtype)

    def run(persist):
        monkeypatch.setenv("MHLIK_CHAIN_PERSIST", persist)
        with Likelihood(N, d, precision=3, segments=segments, timing=True) as lk:
            lk.fill_synthetic(11)
            cp = lk.chain_begin(np.full(d, 0.01, dtype=dtype), 10.0, dtype=dtype)
            outs = [lk.chain_run(inc[:100], logu[:100], dtype=dtype), lk.chain_run(inc[100:101], logu[100:101], dtype=dtype),
                    lk.chain_run(inc[101:], logu[101:], dtype=dtype)]
            return cp, outs, lk.chain_state(dtype=dtype), lk.last_kernel_name()

    cp0, a, st0, name0 = run("0")
    cp1, b, st1, name1 = run("1")
    assert name0.startswith("lik_tile_kernel") and name1.startswith("chain_persist_kernel")
    assert cp0 == cp1
    for x, y in zip(a, b):
        for k in ("samples", "lik", "ratio", "accepted"):
            assert np.array_equal(x[k], y[k]), k
        assert x["n_accepted"] == y["n_accepted"] and y["kernel_ms"] > 0
    assert 0 < sum(x["n_accepted"] for x in a) < n
    assert np.array_equal(st0[0], st1[0]) and st0[1] == st1[1]
