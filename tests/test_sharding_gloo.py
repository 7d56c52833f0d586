"""The N>1 host path on CPU: world_size-2 gloo group, rows sharded by fixed
segments, per-segment sums all-reduced and added in segment order.  The local
evaluator here is a test double backed by the ORACLE (the product's own local
evaluator is the HIP context and needs a GPU: tests/test_gpu_sharded.py)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import oracle_py as O
from mcmc_amd.distributed import combine_segment_sums, segment_rows, shard_rows
from mcmc_amd.likelihood import synthetic_host

N, D, SEED = 30011, 6, 21


class OracleShard:
    """Stands in for mcmc_amd.Likelihood on a machine without a GPU."""

    def __init__(self, n_rows, n_cols, precision=1, rank=0, world=1, segments=8, device=-1, segment_align=0):
        self.segments = segments
        self.begin, self.rows, self.segs = shard_rows(n_rows, rank, world, segments, segment_align or 1)
        self.seg_rows = segment_rows(n_rows, segments, segment_align or 1)
        self.X = synthetic_host(SEED, n_rows, n_cols, self.begin, self.rows)
        self._last = np.zeros(segments)

    def loglik(self, theta):
        self._last = np.zeros(self.segments)
        for g in self.segs:
            b, n = self.seg_rows[g]
            Xs = np.asfortranarray(self.X[b - self.begin:b - self.begin + n])
            self._last[g] = O.log_likelihood(Xs, theta)[0] if n else 0.0
        return combine_segment_sums(self._last)

    def segment_sums(self):
        return self._last.copy()

    def close(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, align):
    import torch.distributed as dist

    from mcmc_amd.distributed import ShardedLikelihood

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sl = ShardedLikelihood(N, D, transport="torch", local_factory=OracleShard, segment_align=align)
    out = []
    rng = np.random.default_rng(0)
    for _ in range(3):
        th = rng.normal(0, 0.3, D)
        out.append(sl.loglik(th))
    q.put((rank, sl.begin, sl.rows, out))
    dist.barrier()
    dist.destroy_process_group()


# world 8 = one segment per rank (the 8-GPU node); align 8192: four segments hold all 30011 rows, ranks 4-7 hold none
@pytest.mark.parametrize("world,align", [(2, 0), (4, 0), (8, 0), (8, 8192)])
def test_sharded_sum_is_world_invariant(world, align):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, align)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the shards tile the rows
    assert got[0][1] == 0 and sum(g[2] for g in got) == N
    for a, b in zip(got, got[1:]):
        assert a[1] + a[2] == b[1]
    if align == 8192:
        assert [g[2] for g in got] == [8192, 8192, 8192, 30011 - 3 * 8192, 0, 0, 0, 0]
    # every rank has the same bits
    for g in got[1:]:
        assert g[3] == got[0][3]
    # and they equal the single-process sum in segment order (world = 1)
    single = OracleShard(N, D, segment_align=align)
    rng = np.random.default_rng(0)
    X = synthetic_host(SEED, N, D)
    for v in got[0][3]:
        th = rng.normal(0, 0.3, D)
        assert single.loglik(th) == v
        whole = O.log_likelihood(X, th)[0]
        assert abs(v - whole) <= 1e-12 * abs(whole)


def test_segment_geometry():
    for n in (0, 1, 7, 8, 9, 10_000_000, 100_000_000):
        rows = segment_rows(n, 8)
        assert sum(r for _, r in rows) == n
        assert all(rows[i][0] + rows[i][1] == rows[i + 1][0] for i in range(7))
        for world in (1, 2, 4, 8):
            for align in (1, 1024):
                arows = segment_rows(n, 8, align)
                tot = 0
                for r in range(world):
                    b, c, segs = shard_rows(n, r, world, 8, align)
                    assert b == arows[segs[0]][0]
                    tot += c
                assert tot == n
    with pytest.raises(ValueError):
        shard_rows(100, 0, 3, 8)
