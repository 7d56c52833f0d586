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


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from mcmc_amd.distributed import ShardedLikelihood

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sl = ShardedLikelihood(N, D, transport="torch", device=0)
    sl.local.fill_synthetic(SEED)
    rng = np.random.default_rng(0)
    out = [sl.loglik(rng.normal(0, 0.05, D)) for _ in range(3)]
    q.put((rank, out))
    dist.barrier()
    sl.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_hip_equals_single_context_bitwise(world):
    from mcmc_amd import Likelihood

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for g in got[1:]:
        assert g[1] == got[0][1]
    rng = np.random.default_rng(0)
    with Likelihood(N, D, device=0) as lk:
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
