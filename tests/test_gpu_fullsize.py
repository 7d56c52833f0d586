"""BASELINE.json's full sizes (N=10M d=64, N=100M d=64, N=10M d=128 K=64), where
the CPU oracle would take minutes per evaluation: parity through properties
that do not depend on the size --
  * L(0) = -N log 2 (every row contributes -log(1+e^0));
  * additivity over row shards: the per-segment sums of rank r of a world-G
    context are, bit for bit, segments [r*S/G, (r+1)*S/G) of the single
    context's (the sum is grouped by fixed segments, not by GPU count);
  * the two HBM layouts hold the same rows in the same tiles: same bits;
  * sampled row dots against the host generator's rows in the gslcblas order:
    bit-exact;
  * SP within 1e-4 of DP; every chain of the batched MFMA passes within its
    tolerance of the single-theta pass of the same precision;
  * a K-chain mixed-precision call = the single-chain call, chain by chain.
The data is generated on the device (mhlik_fill_synthetic); the host generator
produces the same bits for any row range (tests/test_gpu_parity_dp.py)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 12345


def test_dp_sp_at_n10m_d64():
    from mcmc_amd import Likelihood, F64, F32
    from mcmc_amd.likelihood import synthetic_host, LAYOUT_COLMAJOR

    N, d = 10_000_000, 64
    rng = np.random.default_rng(0)
    th = rng.normal(0, 0.02, d)
    with Likelihood(N, d, precision=F64 | F32) as lk:
        lk.fill_synthetic(SEED)
        L0 = lk.loglik(np.zeros(d))
        assert abs(L0 + N * math.log(2.0)) <= 1e-12 * N  # one rounding of log 2 per row at most
        L = lk.loglik(th)
        assert lk.loglik(th) == L  # run-to-run reproducible
        segs = lk.segment_sums()
        assert abs(sum(segs) - L) <= 1e-12 * abs(L)
        sp = lk.loglik_f32(th.astype(np.float32))
        assert abs(sp - L) <= 1e-4 * abs(L)
        # sampled row dots, bit-exact against the gslcblas order on the host rows
        dots = lk.rowdots(th)
        for r0 in (0, 511, 512, 4_999_999, 9_999_488, N - 1):
            x = synthetic_host(SEED, N, d, row0=r0, rows=1)[0]
            acc = 0.0
            for j in range(d):
                acc = acc + th[j] * x[j]
            assert dots[r0] == acc, r0
    # shards of a world-2 / world-8 run: same segment sums, bit for bit
    for world in (2, 8):
        for rank in (0, world - 1):
            with Likelihood(N, d, precision=F64, rank=rank, world=world) as sh:
                sh.fill_synthetic(SEED)
                sh._lib.mhlik_loglik_f64  # no communicator: the call returns this rank's part
                part = _local_eval(sh, th)
                per = 8 // world
                assert list(part[rank * per:(rank + 1) * per]) == list(segs[rank * per:(rank + 1) * per])
    with Likelihood(N, d, precision=F64, layout=LAYOUT_COLMAJOR) as cm:
        cm.fill_synthetic(SEED)
        assert cm.loglik(th) == L


def _local_eval(lk, th):
    """Evaluate without a communicator: the slots of the other ranks stay zero."""
    lk.loglik(th)
    return lk.segment_sums()


def test_dp_at_n100m_d64():
    from mcmc_amd import Likelihood, F64

    N, d = 100_000_000, 64
    th = np.random.default_rng(1).normal(0, 0.01, d)
    with Likelihood(N, d, precision=F64) as lk:  # 51.2 GB
        lk.fill_synthetic(SEED)
        L0 = lk.loglik(np.zeros(d))
        assert abs(L0 + N * math.log(2.0)) <= 1e-12 * N
        L = lk.loglik(th)
        segs = lk.segment_sums()
        info = lk.info()
        assert info["local_rows"] == N and info["hbm_bytes"] >= N * d * 8
    # one of 8 shards (6.4 GB): its segment is the whole job's segment, bit for bit
    with Likelihood(N, d, precision=F64, rank=5, world=8) as sh:
        sh.fill_synthetic(SEED)
        part = _local_eval(sh, th)
        assert part[5] == segs[5]
    assert abs(sum(segs) - L) <= 1e-12 * abs(L)


def test_batched_at_n10m_d128_k64():
    from mcmc_amd import Likelihood, F64, F32

    N, d, K = 10_000_000, 128, 64
    Th = np.random.default_rng(2).normal(0, 0.01, (K, d))
    with Likelihood(N, d, precision=F64 | F32) as lk:
        lk.fill_synthetic(SEED)
        b64 = lk.loglik_batched(Th, dtype=np.float64)
        b32 = lk.loglik_batched(Th.astype(np.float32), dtype=np.float32)
        for k in (0, 17, 63):
            one = lk.loglik(Th[k])
            assert abs(b64[k] - one) <= 1e-10 * abs(one), k
            onef = lk.loglik_f32(Th[k].astype(np.float32))
            assert abs(b32[k] - onef) <= 2e-6 * abs(onef), k
        assert np.max(np.abs(b32 - b64) / np.abs(b64)) <= 1e-4


def test_mp_batched_at_n10m_d128():
    from mcmc_amd import Likelihood

    N, d, K, F, eps = 10_000_000, 128, 3, 16, 1.7e-7
    blocks = N // 64
    rng = np.random.default_rng(3)
    Ths = [rng.normal(0, 0.01, (K, d)) for _ in range(3)]
    chooses = [rng.integers(0, F, K).astype(np.int32) for _ in range(3)]
    seeds = np.full(K, 1234, dtype=np.uint64)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.fill_synthetic(SEED)
        lk.mp_reset_batched(64, blocks, K)
        got = [lk.mp_loglik_batched(Ths[c], F, chooses[c], seeds, c, eps) for c in range(3)]
        for k in (0, K - 1):
            lk.mp_reset(64, blocks)
            for c in range(3):
                v, nb, nd = lk.mp_loglik(Ths[c][k], 64, blocks, F, int(chooses[c][k]), 1234, c, eps)
                assert (got[c][1][k], got[c][2][k]) == (nb, nd), (c, k)
                assert abs(got[c][0][k] - v) <= 2e-6 * abs(v), (c, k)
        # against double precision: the augmented likelihood is within the float pass's error of it
        dp = lk.loglik_batched(Ths[2], dtype=np.float64)
        assert np.max(np.abs(got[2][0] - dp) / np.abs(dp)) <= 1e-4


@pytest.mark.parametrize("f32_mode", ["split", "exact"])
def test_mp_batched_k64_shared_phase_at_n10m_d128(f32_mode):
    """BASELINE configs[4] at its full shape, the way `bench.py --workload mp_batched` (and extra.kchain of the default
    line) runs it: K = 64 chains on ONE shared re-examination phase, F = 64 -- i.e. mpb_dark_x3_kernel (split) /
    mpb_dark_kernel (exact) + mpb_fix_dense_kernel + mpb_final_kernel.  Three calls (the indicators evolve); chains
    0, 31 and 63 against the single-chain mhlik_mp_loglik sequence (the definition of mcmc_mp.cu:475-552 per chain:
    counts identical, value within 2e-6) and the last call against the double-precision K-chain pass (1e-4)."""
    from mcmc_amd import Likelihood

    N, d, K, F, eps = 10_000_000, 128, 64, 64, 1.7e-7
    blocks = N // 64
    rng = np.random.default_rng(4)
    Ths = [rng.normal(0, 0.01, (K, d)) for _ in range(3)]
    phases = [int(rng.integers(0, F)) for _ in range(3)]
    seeds = (1234 + np.arange(K)).astype(np.uint64)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.fill_synthetic(SEED)
        lk.set_batched_f32_mode(f32_mode)
        lk.mp_reset_batched(64, blocks, K)
        got = [lk.mp_loglik_batched(Ths[c], F, np.full(K, phases[c], dtype=np.int32), seeds, c, eps) for c in range(3)]
        assert np.all(got[0][1] + got[0][2] == blocks)
        assert np.all(got[2][1] < blocks // 100)  # steady state: nearly every block dark
        for k in (0, 31, 63):
            lk.mp_reset(64, blocks)
            for c in range(3):
                v, nb, nd = lk.mp_loglik(Ths[c][k], 64, blocks, F, phases[c], int(seeds[k]), c, eps)
                assert (got[c][1][k], got[c][2][k]) == (nb, nd), (c, k)
                assert abs(got[c][0][k] - v) <= 2e-6 * abs(v), (c, k)
        dp = lk.loglik_batched(Ths[2], dtype=np.float64)
        assert np.max(np.abs(got[2][0] - dp) / np.abs(dp)) <= 1e-4
