"""Row-sharded likelihood over several GPUs, one process per GPU.

The reference is single-GPU (cudaSetDevice(0) everywhere, mcmc_gpu.cu:28); the
sharding is new.  L(theta) is a sum over rows, so the N rows are cut into
`segments` fixed, contiguous row segments (8 by default); rank r of `world`
holds segments [r*segments/world, (r+1)*segments/world).  Each evaluation
produces one partial sum per segment; the per-segment sums of all ranks are
all-reduced (a rank contributes zeros for segments it does not hold, so the
all-reduce is exact) and added in segment order.  The result therefore has
the same bits for every world size that divides `segments`, on every rank.

Two transports:
  "rccl"  -- the all-reduce runs inside libmhlik.so on the context's stream
             (ncclAllReduce over xGMI); torch.distributed is used once, to
             ship the 128-byte communicator id.  This is the bench path.
  "torch" -- the segment sums go through torch.distributed.all_reduce on
             whatever backend the group uses (gloo in the CPU tests).
"""
import numpy as np


def segment_rows(n_rows, segments=8, align=1):
    """[(begin, rows)] of the fixed row segments (same rule as mhlik_create; `align` = config.segment_align)."""
    per = (n_rows + segments - 1) // segments
    if align > 1:
        per = (per + align - 1) // align * align
    out = []
    for g in range(segments):
        b = min(g * per, n_rows)
        e = min((g + 1) * per, n_rows)
        out.append((b, e - b))
    return out


def shard_rows(n_rows, rank, world, segments=8, align=1):
    """(first global row, row count, [segment indices]) held by `rank`."""
    if segments % world != 0:
        raise ValueError("segments=%d must be a multiple of world=%d" % (segments, world))
    per = segments // world
    segs = list(range(rank * per, (rank + 1) * per))
    rows = segment_rows(n_rows, segments, align)
    begin = rows[segs[0]][0]
    count = sum(rows[g][1] for g in segs)
    return begin, count, segs


def combine_segment_sums(slots):
    """Add the all-reduced per-segment sums in segment order (fixed order =>
    identical bits on every rank and for every world size)."""
    total = 0.0
    for v in slots:
        total += float(v)
    return total


class ShardedLikelihood:
    """L(theta) over rows sharded across the ranks of a torch.distributed group."""

    def __init__(self, n_rows, n_cols, precision=1, segments=8, transport="rccl", group=None, device=None,
                 local_factory=None, **kw):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.segments = segments
        self.transport = transport if self.world > 1 else "none"
        self.begin, self.rows, self.segs = shard_rows(n_rows, self.rank, self.world, segments,
                                                      int(kw.get("segment_align", 0) or 1))
        if local_factory is None:
            from .likelihood import Likelihood

            local_factory = Likelihood
        self.local = local_factory(n_rows, n_cols, precision=precision, rank=self.rank, world=self.world,
                                   segments=segments, device=-1 if device is None else device, **kw)
        if self.transport == "rccl":
            self._init_rccl()

    def _init_rccl(self):
        """Rank 0 makes the RCCL id; torch.distributed ships the bytes."""
        from .likelihood import Likelihood

        box = [Likelihood.make_comm_id() if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        self.local.comm_init(box[0])

    def _reduce_slots(self, slots):
        import torch

        t = torch.from_numpy(np.ascontiguousarray(slots, dtype=np.float64))
        backend = self.dist.get_backend(self.group)
        if backend == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def loglik(self, theta):
        v = self.local.loglik(theta)
        if self.transport in ("rccl", "none"):
            return v  # already the sum over all shards
        slots = self._reduce_slots(self.local.segment_sums())
        return combine_segment_sums(slots)

    # -- mixed precision over the shards (needs segment_align = a multiple of 1024) --
    def mp_reset(self, block_rows, blocks):
        self.local.mp_reset(block_rows, blocks)

    def mp_loglik(self, theta, block_rows, blocks, resamp_factor, resamp_choose, u_seed, u_call, bound_eps):
        """(value, bright blocks, dark blocks) over all shards: every rank evaluates the data blocks of its own rows;
        the per-segment sums are all-reduced and added in segment order (mhlik_mp_loglik's own combination)."""
        v, nb, nd = self.local.mp_loglik(theta, block_rows, blocks, resamp_factor, resamp_choose, u_seed, u_call, bound_eps)
        if self.transport in ("rccl", "none"):
            return v, nb, nd
        slots = self._reduce_slots(self.local.mp_last_slots())
        S = self.segments
        red_d = red_f = bright = 0.0
        for g in range(S):
            red_d += float(slots[g])
            red_f += float(slots[S + g])
            bright += float(slots[2 * S + g])
        return red_d + float(np.float32(red_f)), int(bright), int(blocks) - int(bright)

    def mp_reset_batched(self, block_rows, blocks, K):
        self.local.mp_reset_batched(block_rows, blocks, K)

    def mp_loglik_batched(self, Theta, resamp_factor, resamp_choose, u_seed, u_call, bound_eps):
        """(values[K], bright[K], dark[K]) of K chains over all shards (mhlik_mp_loglik_batched's own combination:
        per chain the segments' sums in segment order, the float sums rounded to float once)."""
        v, nb, nd = self.local.mp_loglik_batched(Theta, resamp_factor, resamp_choose, u_seed, u_call, bound_eps)
        if self.transport in ("rccl", "none"):
            return v, nb, nd
        slots = self._reduce_slots(self.local.mp_last_slots_batched().ravel()).reshape(len(v), 4, self.segments)
        blocks = self.local._mpb[1]
        out = np.empty(len(v))
        bright = np.empty(len(v), dtype=np.int32)
        for k in range(len(v)):
            red_d = red_f = b = 0.0
            for g in range(self.segments):
                red_d += float(slots[k, 0, g])
                red_f += float(slots[k, 1, g])
                b += float(slots[k, 2, g])
            out[k] = red_d + float(np.float32(red_f))
            bright[k] = int(b)
        return out, bright, (blocks - bright).astype(np.int32)

    def close(self):
        self.local.close()
