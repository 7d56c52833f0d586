"""The HIP path against the COMMITTED fixtures (tests/golden/oracle_golden.json): nothing from oracle/ runs here,
so these hold even if the oracle library and the product drifted together.  The fixtures pin the CPU restatement
(the reference commits no outputs: "parity unpinned" against a reference binary).
Bars: row dots bit-exact; log-likelihood within 1e-10 relative; accept/reject sequence, log u and the last sample
identical; ratio within 1e-10 relative of its likelihood scale."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.json")))


def unhex(v):
    return np.array([float.fromhex(x) for x in v])


def test_golden_rowdots_are_bit_exact():
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    g = GOLD["rowdots"]
    X = synthetic_host(g["data_seed"], g["N"], g["d"])
    with Likelihood(g["N"], g["d"]) as lk:
        lk.upload(X)
        dots = lk.rowdots(unhex(g["theta"]))
        val = lk.loglik(unhex(g["theta"]))
    assert np.array_equal(dots, unhex(g["mvout"]))
    want = float.fromhex(g["loglik"])
    assert abs(val - want) <= 1e-10 * abs(want)


def test_golden_edge_cases():
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    for e in GOLD["edges"]:
        if e["name"] == "overflow":
            X = np.zeros((1000, 2), order="F")
            X[:, 0] = 1.0
            X[:, 1] = np.linspace(-1, 1, 1000)
        else:
            X = synthetic_host(e["data_seed"], e["N"], e["d"])
        th = unhex(e["theta"])
        with Likelihood(e["N"], e["d"]) as lk:
            lk.upload(X)
            val = lk.loglik(th)
            batched = lk.loglik_batched(np.stack([th, th]))
        want = float.fromhex(e["loglik"])
        if np.isinf(want):
            assert val == want and batched[0] == want, e
        else:
            assert abs(val - want) <= 1e-10 * abs(want), (e["name"], val, want)
            assert abs(batched[1] - want) <= 1e-10 * abs(want), (e["name"], batched, want)


@pytest.mark.parametrize("idx", range(len(GOLD["chains"])))
def test_golden_chains(idx):
    """gpu_sampler (device-resident chain) and the K-chain sampler reproduce the frozen traces."""
    import hashlib

    from mcmc_amd import Likelihood, samplers as S
    from mcmc_amd.likelihood import synthetic_host

    ch = GOLD["chains"][idx]
    X = synthetic_host(ch["data_seed"], ch["N"], ch["d"])
    assert hashlib.sha256(np.ascontiguousarray(X).tobytes()).hexdigest() == ch["x_sha256"], "data generator changed"
    rwsd = float.fromhex(ch["rwsd"])
    run = S.gpu_sampler(X, ch["chain_seed"], ch["Ns"], ch["burnin"], rwsd)
    tr = run["trace"]
    lik = unhex(ch["lik"])
    assert [int(v) for v in tr["accept"]] == ch["accept"]
    assert np.array_equal(tr["logu"], unhex(ch["logu"]))
    assert np.all(np.abs(tr["lik"] - lik) <= 1e-10 * np.abs(lik))
    assert np.all(np.abs(tr["ratio"] - unhex(ch["ratio"])) <= 1e-10 * np.abs(lik))
    assert np.array_equal(run["samples"][-1], unhex(ch["last_sample"]))
    assert run["acceptance"] == float.fromhex(ch["acceptance"])
    with Likelihood(ch["N"], ch["d"]) as lk:
        lk.upload(X)
        multi = S.multichain_sampler(lk, [ch["chain_seed"]], ch["Ns"], ch["burnin"], rwsd)
    assert np.array_equal(multi["samples"][0][-1], unhex(ch["last_sample"]))
