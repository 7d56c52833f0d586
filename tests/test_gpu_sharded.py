"""Row sharding with the REAL HIP kernels: two processes on the one GPU of the
box, each holding half of the rows (world=2), per-segment sums combined over
a gloo group.  The result must have the same bits as the single-context run
(the sum is grouped by fixed segments, not by GPU count).  The RCCL transport
needs two GPUs and is exercised by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N, D, SEED = 700_001, 16, 12345


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, n_rows, align):
    import torch.distributed as dist

    from mcmc_amd.distributed import ShardedLikelihood

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sl = ShardedLikelihood(n_rows, D, transport="torch", device=0, segment_align=align)
    sl.local.fill_synthetic(SEED)
    rng = np.random.default_rng(0)
    out = [sl.loglik(rng.normal(0, 0.05, D)) for _ in range(3)]
    q.put((rank, out, sl.local.info()["local_rows"]))
    dist.barrier()
    sl.close()
    dist.destroy_process_group()


def _spawn(target, world, args):
    """`world` processes on the one GPU of the box (the parent holds no context meanwhile); their queue items sorted
    by rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


# world 8 is the target machine's shape: ONE segment per rank.  (8, 5000, 1024): segments rounded up to 1024 rows
# cover the 5000 rows with five of them, so ranks 5, 6 and 7 hold no rows at all.
@pytest.mark.parametrize("world,n_rows,align", [(2, N, 0), (4, N, 0), (8, N, 0), (8, 5000, 1024), (8, 9, 0)])
def test_sharded_hip_equals_single_context_bitwise(world, n_rows, align):
    from mcmc_amd import Likelihood

    got = _spawn(_worker, world, (n_rows, align))
    for g in got[1:]:
        assert g[1] == got[0][1]
    assert sum(g[2] for g in got) == n_rows
    if (n_rows, align) == (5000, 1024):
        assert [g[2] for g in got] == [1024, 1024, 1024, 1024, 904, 0, 0, 0]
    rng = np.random.default_rng(0)
    with Likelihood(n_rows, D, device=0, segment_align=align) as lk:
        lk.fill_synthetic(SEED)
        for v in got[0][1]:
            assert lk.loglik(rng.normal(0, 0.05, D)) == v


def test_rccl_path_with_one_rank(monkeypatch):
    """The in-library RCCL transport on the one GPU of the box: a 1-rank
    communicator (MHLIK_FORCE_COMM=1) runs the same code as N ranks do --
    dlopen of librccl, id, ncclCommInitRank, ncclAllReduce on the context's
    stream, publish kernel -- and must not change a bit of the result."""
    from mcmc_amd import Likelihood

    n, d = 300_000, 16
    rng = np.random.default_rng(1)
    ths = [rng.normal(0, 0.05, d) for _ in range(3)]
    with Likelihood(n, d, device=0) as lk:
        lk.fill_synthetic(SEED)
        plain = [lk.loglik(t) for t in ths]
        plain_b = lk.loglik_batched(np.array(ths))
    monkeypatch.setenv("MHLIK_FORCE_COMM", "1")
    with Likelihood(n, d, device=0) as lk:
        lk.fill_synthetic(SEED)
        lk.comm_init(Likelihood.make_comm_id())
        assert [lk.loglik(t) for t in ths] == plain
        assert np.array_equal(lk.loglik_batched(np.array(ths)), plain_b)


# ---- mixed precision over row shards ------------------------------------------------------------------------------
MP_N, MP_D, MP_BLOCK, MP_RF, MP_EPS = 300_007, 6, 128, 4, 1.7e-7
MP_BLOCKS = (MP_N + MP_BLOCK - 1) // MP_BLOCK + 3  # = _mp_blocks(MP_N)


def _mp_thetas():
    rng = np.random.default_rng(8)
    return [rng.normal(0, 0.4, MP_D) for _ in range(5)]


def _mp_blocks(n_rows):
    return (n_rows + MP_BLOCK - 1) // MP_BLOCK + 3  # the last block is ragged and three lie beyond the data


def _mp_worker(rank, world, port, q, n_rows):
    import torch.distributed as dist

    from mcmc_amd.distributed import ShardedLikelihood

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blocks = _mp_blocks(n_rows)
    sl = ShardedLikelihood(n_rows, MP_D, precision=3, transport="torch", device=0, segment_align=1024)
    sl.local.fill_synthetic(SEED)
    sl.mp_reset(MP_BLOCK, blocks)
    out = [sl.mp_loglik(th, MP_BLOCK, blocks, MP_RF, call % MP_RF, 1234, call, MP_EPS)
           for call, th in enumerate(_mp_thetas())]
    q.put((rank, out))
    dist.barrier()
    sl.close()
    dist.destroy_process_group()


# (8, 5000): five 1024-row segments hold the rows, ranks 5-7 hold none (and the blocks beyond the data go to rank 7)
@pytest.mark.parametrize("world,n_rows", [(2, MP_N), (4, MP_N), (8, MP_N), (8, 5000)])
def test_sharded_mixed_precision_equals_single_context_bitwise(world, n_rows):
    """mhlik_mp_loglik over row shards (the reference's mixed-precision path is single-GPU, mcmc_mp.cu:286): with
    block-aligned segments every rank evaluates the data blocks of its own rows, and values, bright and dark counts of
    a whole call sequence -- the indicators evolve from call to call -- are those of the single-context run, bit for
    bit; the single-context run is tied to the oracle."""
    import ctypes as C

    import oracle_py as O
    from mcmc_amd import Likelihood
    from mcmc_amd.likelihood import synthetic_host

    got = _spawn(_mp_worker, world, (n_rows,))
    for g in got[1:]:
        assert g[1] == got[0][1]
    blocks = _mp_blocks(n_rows)
    X = synthetic_host(SEED, n_rows, MP_D)
    Xf = np.asfortranarray(X.astype(np.float32))
    L = O.lib()
    L.oracle_mp_likelihood.restype = C.c_double
    L.oracle_mp_likelihood.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    z = np.ones(blocks, dtype=np.int8)
    with Likelihood(n_rows, MP_D, precision=3, device=0, segment_align=1024) as lk:  # 8 segments, one GPU
        lk.fill_synthetic(SEED)
        lk.mp_reset(MP_BLOCK, blocks)
        for call, th in enumerate(_mp_thetas()):
            one = lk.mp_loglik(th, MP_BLOCK, blocks, MP_RF, call % MP_RF, 1234, call, MP_EPS)
            assert one == got[0][1][call], (call, one, got[0][1][call])
            nb, nd = C.c_int(), C.c_int()
            want = L.oracle_mp_likelihood(O.ptr(X), O.ptr(Xf), n_rows, MP_D, n_rows, O.ptr(th), MP_BLOCK, blocks, MP_RF,
                                          call % MP_RF, 1234, call, MP_EPS, O.ptr(z), C.byref(nb), C.byref(nd))
            assert (one[1], one[2]) == (nb.value, nd.value), (call, one, nb.value, nd.value)
            assert abs(one[0] - want) <= 2e-6 * abs(want)
        s_all, e_all = lk.mp_bound(_mp_thetas()[0])
    eps = C.c_double()
    L.oracle_mp_bound.restype = C.c_double
    L.oracle_mp_bound.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.POINTER(C.c_double)]
    want_s = L.oracle_mp_bound(O.ptr(X), O.ptr(Xf), n_rows, MP_D, n_rows, O.ptr(_mp_thetas()[0]), C.byref(eps))
    assert abs(s_all - want_s) <= 1e-10 * abs(want_s) and abs(e_all - eps.value) <= 1e-9


def test_mixed_precision_needs_aligned_segments_and_rccl_path_with_one_rank(monkeypatch):
    from mcmc_amd import Likelihood
    from mcmc_amd._lib import MhlikError

    th = _mp_thetas()[0]
    with Likelihood(MP_N, MP_D, precision=3, device=0) as lk:  # 8 segments of 37501 rows: blocks would straddle them
        lk.fill_synthetic(SEED)
        lk.mp_reset(MP_BLOCK, MP_BLOCKS)
        with pytest.raises(MhlikError, match="segment_align"):
            lk.mp_loglik(th, MP_BLOCK, MP_BLOCKS, MP_RF, 0, 1234, 0, MP_EPS)
    with Likelihood(MP_N, MP_D, precision=3, device=0, segment_align=1024) as lk:
        lk.fill_synthetic(SEED)
        lk.mp_reset(MP_BLOCK, MP_BLOCKS)
        plain = [lk.mp_loglik(t, MP_BLOCK, MP_BLOCKS, MP_RF, i % MP_RF, 1234, i, MP_EPS) for i, t in enumerate(_mp_thetas())]
        plain_bound = lk.mp_bound(th)
    monkeypatch.setenv("MHLIK_FORCE_COMM", "1")  # the in-stream all-reduce of the 3*segments slots, one rank
    with Likelihood(MP_N, MP_D, precision=3, device=0, segment_align=1024) as lk:
        lk.fill_synthetic(SEED)
        lk.comm_init(Likelihood.make_comm_id())
        lk.mp_reset(MP_BLOCK, MP_BLOCKS)
        assert [lk.mp_loglik(t, MP_BLOCK, MP_BLOCKS, MP_RF, i % MP_RF, 1234, i, MP_EPS)
                for i, t in enumerate(_mp_thetas())] == plain
        assert lk.mp_bound(th) == plain_bound


# ---- K-chain mixed precision over row shards ------------------------------------------------------------------------
KMP_K = 21


def _kmp_inputs():
    rng = np.random.default_rng(12)
    thetas = rng.normal(0, 0.4, (5, KMP_K, MP_D))
    phases = rng.integers(0, MP_RF, (5, KMP_K))
    phases[3:] = phases[3:, :1]  # the last two calls on one phase for all chains: the dense correction kernel
    seeds = (500 + np.arange(KMP_K)).astype(np.uint64)
    return thetas, phases, seeds


def _kmp_run(obj, n_rows=MP_N):
    thetas, phases, seeds = _kmp_inputs()
    obj.mp_reset_batched(MP_BLOCK, _mp_blocks(n_rows), KMP_K)
    out = []
    for call in range(len(thetas)):
        v, nb, nd = obj.mp_loglik_batched(thetas[call], MP_RF, phases[call], seeds, call, MP_EPS)
        out.append((v.tolist(), nb.tolist(), nd.tolist()))
    return out


def _kmp_worker(rank, world, port, q, f32_mode, n_rows):
    import torch.distributed as dist

    from mcmc_amd.distributed import ShardedLikelihood

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sl = ShardedLikelihood(n_rows, MP_D, precision=3, transport="torch", device=0, segment_align=1024)
    sl.local.fill_synthetic(SEED)
    sl.local.set_batched_f32_mode(f32_mode)
    q.put((rank, _kmp_run(sl, n_rows)))
    dist.barrier()
    sl.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,f32_mode,n_rows", [(2, "split", MP_N), (4, "exact", MP_N), (8, "split", MP_N),
                                                    (8, "exact", 5000)])
def test_sharded_kchain_mixed_precision_equals_single_context_bitwise(world, f32_mode, n_rows, monkeypatch):
    """mhlik_mp_loglik_batched over row shards: every rank corrects and resamples the data blocks of its own rows
    (per-chain phases and one shared phase, i.e. both correction kernels), the K x 4 x segments slots are summed over
    the ranks, and values, bright and dark counts of a call sequence equal the single-context run with the same
    eight segments bit for bit -- which in turn matches the one-segment path of tests/test_gpu_mp_batched.py in its
    indicators exactly and in its values to float-sum tolerance, and runs unchanged over a forced one-rank RCCL
    communicator (in-stream all-reduce + mpb_combine_kernel).  World 8 = one segment per rank; with 5000 rows the
    last three ranks hold none."""
    from mcmc_amd import Likelihood

    got = _spawn(_kmp_worker, world, (f32_mode, n_rows))
    for g in got[1:]:
        assert g[1] == got[0][1]

    def single(segments, align, force=False):
        with Likelihood(n_rows, MP_D, precision=3, device=0, segments=segments, segment_align=align) as lk:
            lk.fill_synthetic(SEED)
            lk.set_batched_f32_mode(f32_mode)
            if force:
                lk.comm_init(Likelihood.make_comm_id())
            return _kmp_run(lk, n_rows)

    eight = single(8, 1024)
    assert eight == got[0][1]
    one = single(1, 0)
    for (v8, b8, d8), (v1, b1, d1) in zip(eight, one):
        assert b8 == b1 and d8 == d1  # the indicators do not depend on how the sums are cut
        assert np.allclose(v8, v1, rtol=2e-6, atol=0)
    monkeypatch.setenv("MHLIK_FORCE_COMM", "1")
    assert single(8, 1024, force=True) == eight
    assert single(1, 0, force=True) == one  # one segment through the slot path: the same sums


def test_sharded_device_loops_refuse_to_run_without_a_communicator():
    """A context that holds only part of the segments and has no RCCL communicator (gloo transport, or after
    mhlik_comm_destroy) cannot run the device-resident loops: their accept tests would see this rank's share of the
    likelihood only.  Both loops say so instead of letting the ranks' chains part."""
    from mcmc_amd import Likelihood
    from mcmc_amd._lib import MhlikError

    with Likelihood(MP_N, MP_D, precision=3, device=0, rank=1, world=2, segment_align=1024) as lk:
        lk.fill_synthetic(SEED)
        assert lk.loglik(np.zeros(MP_D)) < 0  # the blocking call hands out the rank's share, as documented
        with pytest.raises(MhlikError, match="needs mhlik_comm_init"):
            lk.chain_begin(np.zeros(MP_D), 10.0)
        with pytest.raises(MhlikError, match="needs mhlik_comm_init"):
            lk.chain_resume(np.zeros(MP_D), -1.0, 10.0)
        with pytest.raises(MhlikError, match="needs mhlik_comm_init"):
            lk.kchain_begin(np.zeros((4, MP_D)), 10.0)
        with pytest.raises(MhlikError, match="needs mhlik_comm_init"):
            lk.kchain_resume(np.zeros((4, MP_D)), np.zeros(4), 10.0, mode=2)


def test_comm_destroy_leaves_no_stale_slots(monkeypatch):
    """mhlik_comm_destroy: later evaluations are this context's rows again, with nothing left over from the last
    reduced call in the slot buffers (one-rank communicator: all that a one-GPU box can build)."""
    from mcmc_amd import Likelihood

    monkeypatch.setenv("MHLIK_FORCE_COMM", "1")
    th = np.full(D, 0.01)
    with Likelihood(100_000, D, device=0) as lk:
        lk.fill_synthetic(SEED)
        lk.comm_init(Likelihood.make_comm_id())
        assert lk.comm_ranks() == 1
        reduced = lk.loglik(th)
        lk.comm_destroy()
        assert lk.comm_ranks() == 0 and not lk.segment_sums().any()
        assert lk.loglik(th) == reduced and lk.segment_sums().sum() != 0


@pytest.mark.parametrize("N", [5000, 1024, 700])
def test_kchain_mixed_precision_with_empty_trailing_segments(N):
    """Eight block-aligned segments over so few rows that the trailing segments (and, at world > 1, whole ranks) hold
    none: blocks beyond the data are counted once (they stay bright), and the indicator counts equal the one-segment
    run's call by call."""
    from mcmc_amd import Likelihood

    d, K, block_rows, F = 5, 7, 64, 3
    blocks = (N + block_rows - 1) // block_rows + 2
    rng = np.random.default_rng(N)
    thetas = rng.normal(0, 0.4, (4, K, d))
    phases = rng.integers(0, F, (4, K))
    seeds = (900 + np.arange(K)).astype(np.uint64)

    def run(segments, align):
        with Likelihood(N, d, precision=3, device=0, segments=segments, segment_align=align) as lk:
            lk.fill_synthetic(SEED)
            lk.mp_reset_batched(block_rows, blocks, K)
            lk.mp_reset(block_rows, blocks)
            out = []
            for c in range(4):
                v, nb, nd = lk.mp_loglik_batched(thetas[c], F, phases[c], seeds, c, MP_EPS)
                # chain 0 again through the single-chain entry point (its own indicators, same sequence)
                v0, nb0, nd0 = lk.mp_loglik(thetas[c, 0], block_rows, blocks, F, int(phases[c, 0]), int(seeds[0]), c, MP_EPS)
                assert (nb0, nd0) == (nb[0], nd[0]) and abs(v0 - v[0]) <= 2e-5 * abs(v0) + 1e-9
                out.append((v, nb.copy(), nd.copy()))
            return out

    one, eight = run(1, 0), run(8, 1024)
    for (v1, b1, d1), (v8, b8, d8) in zip(one, eight):
        assert np.array_equal(b1, b8) and np.array_equal(d1, d8)
        assert np.all(b8 >= 2) and np.all(b8 + d8 == blocks)
        assert np.allclose(v1, v8, rtol=2e-6, atol=1e-9)
