"""K-chain mixed-precision likelihood (BASELINE.json config 5: mcmc_mp with
batched chains, X.Theta on MFMA) against the CPU restatement of the
single-chain definition (oracle/mp_path.c) applied chain by chain, and against
the single-chain HIP path.  "Parity unpinned" like the rest of the MP path:
the reference's mcmc_mp is work in progress and commits no outputs."""
import ctypes as C

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


def _oracle_mp(X, Xf, th, block_rows, blocks, rf, rc, seed, call, eps, z):
    L = O.lib()
    L.oracle_mp_likelihood.restype = C.c_double
    L.oracle_mp_likelihood.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    nb, nd = C.c_int(), C.c_int()
    N, d = X.shape
    v = L.oracle_mp_likelihood(O.ptr(X), O.ptr(Xf), N, d, N, O.ptr(th), block_rows, blocks, rf, rc, seed, call, eps,
                               O.ptr(z), C.byref(nb), C.byref(nd))
    return v, nb.value, nd.value


CASES = [
    # N, d, block_rows, blocks (None = N // block_rows), K, resamp_factor
    (16384, 8, 64, None, 64, 2),
    (10000, 3, 64, None, 5, 4),      # ragged last tile, K < 32
    (20000, 16, 256, None, 33, 3),   # K just over one 32-chain block
    (8192, 5, 1024, None, 64, 2),
    (5000, 7, 128, 30, 17, 2),       # blocks cover only part of the rows
    (3000, 4, 64, 50, 8, 5),         # more blocks than rows: trailing blocks hold no rows
    (9000, 130, 64, None, 40, 8),    # wide theta, odd tail columns
    (4096, 200, 64, None, 64, 4),    # K*d too large for the LDS copy of Theta: correction kernel reads it from global memory
    (2048, 1, 64, None, 3, 2),       # a single column
    (10007, 256, 256, 40, 64, 2),    # widest theta; the last block is partial down to a partial 64-row segment
    (9000, 5, 128, None, 20, 3),     # two segments per block: a block is two stripes' work
]


# one phase for all chains: with K >= 16 the correction runs as a dense contraction on the matrix cores
# (mpb_fix_dense_kernel) instead of the per-(block, chain) kernel; the definition per chain is the same
@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("mode", ["exact", "split"])  # float dots of the dark pass: f32 MFMA / three bf16 pieces
@pytest.mark.parametrize("N,d,block_rows,blocks,K,rf", CASES + [(70000, 24, 512, None, 64, 3), (6000, 9, 64, 40, 16, 2)])
def test_mp_batched_sequence_matches_oracle_per_chain(N, d, block_rows, blocks, K, rf, mode, shared):
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    X = synthetic_host(77, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    blocks = N // block_rows if blocks is None else blocks
    eps = 1.7e-7
    z = np.ones((K, blocks), dtype=np.int8)
    seeds = np.arange(1000, 1000 + K, dtype=np.uint64)
    rng = np.random.default_rng(1)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.upload(X)
        lk.upload(Xf)
        lk.mp_reset_batched(block_rows, blocks, K)
        lk.set_batched_f32_mode(mode)
        for call in range(5):
            Th = rng.normal(0, 0.5 / np.sqrt(d), (K, d))
            choose = rng.integers(0, rf, K).astype(np.int32)
            if shared:
                choose[:] = choose[0]
            got, gb, gd = lk.mp_loglik_batched(Th, rf, choose, seeds, call, eps)
            for k in range(K):
                want, wb, wd = _oracle_mp(X, Xf, Th[k], block_rows, blocks, rf, int(choose[k]), int(seeds[k]), call,
                                          eps, z[k])
                assert (gb[k], gd[k]) == (wb, wd), (call, k, gb[k], gd[k], wb, wd)
                # at d = 256 the float dot's error exceeds eps for some rows of a bright block: log(L - B) of a negative
                # number, NaN in the definition and here alike
                assert (np.isnan(want) and np.isnan(got[k])) or abs(got[k] - want) <= 2e-6 * abs(want), (call, k, got[k], want)
            assert np.array_equal(lk.mp_indicators_batched(), z), call


def test_mp_batched_chain_equals_single_chain_path():
    """Chain k of the batched call = mhlik_mp_loglik with chain k's state."""
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d, block_rows, K, rf, eps = 32768, 12, 64, 6, 4, 1.7e-7
    blocks = N // block_rows
    X = synthetic_host(5, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    rng = np.random.default_rng(2)
    Ths = [rng.normal(0, 0.2, (K, d)) for _ in range(4)]
    chooses = [rng.integers(0, rf, K).astype(np.int32) for _ in range(4)]
    seeds = np.arange(7, 7 + K, dtype=np.uint64)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.upload(X)
        lk.upload(Xf)
        lk.mp_reset_batched(block_rows, blocks, K)
        batched = [lk.mp_loglik_batched(Ths[c], rf, chooses[c], seeds, c, eps) for c in range(4)]
        for k in range(K):
            lk.mp_reset(block_rows, blocks)
            for c in range(4):
                v, nb, nd = lk.mp_loglik(Ths[c][k], block_rows, blocks, rf, int(chooses[c][k]), int(seeds[k]), c, eps)
                bv, bb, bd = batched[c]
                assert (bb[k], bd[k]) == (nb, nd)
                assert abs(bv[k] - v) <= 2e-6 * abs(v)


@pytest.mark.parametrize("eps", [1.7e-7, 1e-4])
@pytest.mark.parametrize("scale", [0.3, 3.0, 6.0])
def test_mp_batched_dark_terms_product_form_and_its_limits(scale, eps):
    """The split dark pass adds a lane's 16 terms log(sigma - eps) in product form while every dot of the group is
    below min(7, log(2e-4 (1 - eps) / eps)), and element by element otherwise: small, large (|x| up to ~12) and
    overflowing (x > 15.6: sigma < eps, NaN in the definition as well) dots, and an eps for which the limit is 0.69,
    each against the CPU restatement chain by chain and against the exact float mode (per-element terms)."""
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    N, d, block_rows, K, rf = 8192, 6, 64, 16, 2
    blocks = N // block_rows
    X = synthetic_host(31, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    z = np.ones((K, blocks), dtype=np.int8)
    seeds = np.arange(500, 500 + K, dtype=np.uint64)
    rng = np.random.default_rng(8)
    Ths = [rng.normal(0, scale, (K, d)) for _ in range(4)]
    got = {}
    for mode in ("split", "exact"):
        with Likelihood(N, d, precision=3, segments=1) as lk:
            lk.upload(X)
            lk.upload(Xf)
            lk.mp_reset_batched(block_rows, blocks, K)
            lk.set_batched_f32_mode(mode)
            got[mode] = [lk.mp_loglik_batched(Ths[c], rf, np.full(K, c % rf, dtype=np.int32), seeds, c, eps)
                         for c in range(4)]
    nan_seen = False
    for c in range(4):
        for k in range(K):
            want, wb, wd = _oracle_mp(X, Xf, Ths[c][k], block_rows, blocks, rf, c % rf, int(seeds[k]), c, eps, z[k])
            for mode in ("split", "exact"):
                v, nb, nd = got[mode][c]
                assert (nb[k], nd[k]) == (wb, wd), (mode, c, k)
                if np.isnan(want):
                    nan_seen = True
                    assert np.isnan(v[k]), (mode, c, k, v[k])
                else:
                    assert abs(v[k] - want) <= 2e-6 * abs(want), (mode, c, k, v[k], want)
    # the inputs really reach the regimes named above
    xmax = max(float(np.max(X @ Ths[c].T)) for c in range(4))
    assert (xmax < 3.0) if scale == 0.3 else (xmax > 7.0 if scale == 3.0 else xmax > 15.6), xmax
    assert nan_seen or scale < 6.0


@pytest.mark.parametrize("N,d,block_rows", [(400000, 4, 64), (1500000, 3, 256)])
def test_mp_batched_dense_correction_in_rounds_equals_single_chain_path(N, d, block_rows):
    """24 blocks per workgroup of the dense correction kernel: with everything bright (the first call) a workgroup
    takes its visited blocks in two rounds of 16 (64-row blocks) or six of 4 (256-row blocks: four table slots each).
    Checked against the single-chain HIP path rather than the CPU restatement: 10^5 bright (block, chain) decisions
    per call each hang on logf(B), where ROCm's logf and glibc's may differ in the last place (the reference's
    formula, mcmc_mp.cu:148, has the same freedom), and that moves about one decision in 10^5."""
    from mcmc_amd import Likelihood

    K, rf, eps = 16, 2, 1.7e-7
    blocks = N // block_rows
    rng = np.random.default_rng(3)
    Ths = [rng.normal(0, 0.2, (K, d)) for _ in range(3)]
    chooses = [np.full(K, c % rf, dtype=np.int32) for c in range(3)]
    seeds = np.arange(70, 70 + K, dtype=np.uint64)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.fill_synthetic(11)
        lk.mp_reset_batched(block_rows, blocks, K)
        batched = [lk.mp_loglik_batched(Ths[c], rf, chooses[c], seeds, c, eps) for c in range(3)]
        for k in (0, 7, 15):
            lk.mp_reset(block_rows, blocks)
            for c in range(3):
                v, nb, nd = lk.mp_loglik(Ths[c][k], block_rows, blocks, rf, int(chooses[c][k]), int(seeds[k]), c, eps)
                bv, bb, bd = batched[c]
                assert (bb[k], bd[k]) == (nb, nd), (k, c)
                assert abs(bv[k] - v) <= 2e-6 * abs(v), (k, c, bv[k], v)


def test_mp_batched_rejects_bad_arguments():
    from mcmc_amd import Likelihood
    from mcmc_amd._lib import MhlikError
    from mcmc_amd.likelihood import synthetic_host

    N, d = 4096, 3
    X = synthetic_host(1, N, d)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.upload(X)
        lk.upload(np.asfortranarray(X.astype(np.float32)))
        with pytest.raises(MhlikError):
            lk.mp_reset_batched(96, 10, 4)  # not a power of two
        with pytest.raises(MhlikError):
            lk.mp_reset_batched(64, 10, 65)  # too many chains
        lk.mp_reset_batched(64, N // 64, 4)
        with pytest.raises(MhlikError):  # resamp_choose outside [0, factor)
            lk.mp_loglik_batched(np.zeros((4, d)), 2, [0, 1, 2, 0], [1, 2, 3, 4], 0, 1e-7)


@pytest.mark.parametrize("mode", ["exact", "split"])
def test_mp_multichain_sampler_reproduces_mp_sampler_per_chain(mode):
    """Chain k of the lock-step sampler walks the chain mp_sampler walks with
    the same generator seed: same proposals, same accept decisions, same
    samples (the dark float sums agree to ~1e-7 relative, far inside the
    accept margins of this short run)."""
    from mcmc_amd import Likelihood
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d, Ns, burnin = 8192, 4, 60, 20
    X = synthetic_host(12345, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    seeds = [3, 4, 3]
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.upload(X)
        lk.upload(Xf)
        lk.set_batched_f32_mode(mode)
        out = S.mp_multichain_sampler(lk, seeds, Ns, burnin, 0.05, 64, N // 64, 4, 1.7e-7)
    assert np.isfinite(out["samples"]).all()
    assert np.array_equal(out["samples"][0], out["samples"][2])  # same seed, same chain
    assert not np.array_equal(out["samples"][0], out["samples"][1])
    for k in (0, 1):
        ref = S.mp_sampler(X, seeds[k], Ns, burnin, 0.05, block_rows=64, blocks=N // 64, resamp_factor=4, bound=1.7e-7,
                           trace=False)
        assert np.array_equal(out["samples"][k], ref["samples"]), k
        assert np.array_equal(out["burn"][k][1:], ref["burn"][1:]), k
        assert out["accepted"][k] == round(ref["acceptance"] * Ns)


def test_mp_multichain_sampler_shared_phase_runs_and_keeps_streams():
    from mcmc_amd import Likelihood
    from mcmc_amd import samplers as S
    from mcmc_amd.likelihood import synthetic_host

    N, d = 8192, 4
    X = synthetic_host(12345, N, d)
    with Likelihood(N, d, precision=3, segments=1) as lk:
        lk.upload(X)
        lk.upload(np.asfortranarray(X.astype(np.float32)))
        a = S.mp_multichain_sampler(lk, [3, 4, 5], 40, 10, 0.05, 64, N // 64, 8, 1.7e-7, shared_phase=True)
        b = S.mp_multichain_sampler(lk, [3, 4, 5], 40, 10, 0.05, 64, N // 64, 8, 1.7e-7, shared_phase=False)
    assert np.isfinite(a["samples"]).all()
    # chain 0 re-examines its own phase either way: identical trajectory
    assert np.array_equal(a["samples"][0], b["samples"][0])
    assert (a["bright"] >= 0).all() and (a["bright"] <= N // 64).all()


def test_mfma_16x16x4_f32_is_the_column_ordered_fmaf_chain():
    """mpb_fix_dense_kernel forms the float dots of bright / re-examined rows with v_mfma_f32_16x16x4_f32 and
    relies on it being, bit for bit, the k-ordered fmaf chain of the single-chain kernel with the result rows
    4*(lane>>4)+r in a lane (the double MFMA: rows (lane>>4)+4*r): tools/check_mfma_order."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_mfma_order")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert "row=4*(l>>4)+r: f32 16x16x4 == k-ordered fmaf chain on 256/256" in lines[0], out.stdout
    assert "f64 16x16x4 == fma chain on 256/256" in lines[1], out.stdout


def test_mp_batched_begin_end_equals_the_blocking_call():
    """mhlik_mp_loglik_batched_begin + _end on one context against the blocking call on a twin context (the
    indicators evolve with every call, so the two run side by side); no other evaluation between the halves."""
    import ctypes as C

    from mcmc_amd import Likelihood, _lib as L
    from mcmc_amd._lib import MhlikError
    from mcmc_amd.likelihood import synthetic_host

    N, d, K, rf, eps = 16384, 6, 20, 4, 1.7e-7
    blocks = N // 64
    X = synthetic_host(11, N, d)
    Xf = np.asfortranarray(X.astype(np.float32))
    seeds = np.arange(50, 50 + K, dtype=np.uint64)
    rng = np.random.default_rng(8)
    with Likelihood(N, d, precision=3, segments=1) as a, Likelihood(N, d, precision=3, segments=1) as b:
        for lk in (a, b):
            lk.upload(X)
            lk.upload(Xf)
            lk.mp_reset_batched(64, blocks, K)
        lib = L.load_mhlik()
        for call in range(4):
            Th = np.ascontiguousarray(rng.normal(0, 0.2, (K, d)))
            choose = np.full(K, rng.integers(0, rf), dtype=np.int32) if call % 2 else rng.integers(0, rf, K).astype(np.int32)
            want, wb, wd = a.mp_loglik_batched(Th, rf, choose, seeds, call, eps)
            st = L.MpBatch(64, blocks, rf, K, choose.ctypes.data, seeds.ctypes.data, call, eps)
            assert lib.mhlik_mp_loglik_batched_begin(b.handle, C.byref(st), Th.ctypes.data) == 0
            with pytest.raises(MhlikError):
                b.loglik(Th[0])  # the context is busy until _end
            with pytest.raises(MhlikError):
                b.mp_indicators_batched()
            got, nb, nd = np.empty(K), np.empty(K, dtype=np.int32), np.empty(K, dtype=np.int32)
            assert lib.mhlik_mp_loglik_batched_end(b.handle, got.ctypes.data, nb.ctypes.data, nd.ctypes.data) == 0
            assert np.array_equal(got, want) and np.array_equal(nb, wb) and np.array_equal(nd, wd)
            assert np.array_equal(a.mp_indicators_batched(), b.mp_indicators_batched())
        assert lib.mhlik_mp_loglik_batched_end(b.handle, got.ctypes.data, None, None) != 0  # nothing begun
