"""Row sharding over the GPUs of one node: the n rows (and their labels) are split
contiguously over the ranks, the M centres, the preconditioner and every CG vector are
replicated, and each CG iteration exchanges exactly one all-reduce(sum) of the (M,) f64
partial of K_nM'(K_nM v).  One process per GPU; the collective is RCCL through
torch.distributed (backend "nccl" on ROCm) — "gloo" in the CPU tests.

The reference has no counterpart (it is single-process); this is new design.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous, balanced split of n rows: the first n % world ranks get one extra row."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class RowShard:
    def __init__(self, group=None):
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0

    def bounds(self, n):
        return shard_bounds(n, self.world, self.rank)

    def allreduce(self, v):
        """In-place sum over the row shards; returns v."""
        if self.enabled:
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
        return v

    def total(self, n_local):
        if not self.enabled:
            return int(n_local)
        t = torch.tensor([int(n_local)], dtype=torch.int64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def broadcast(self, t, src=0):
        if self.enabled:
            dist.broadcast(t, src=src, group=self.group)
        return t

    # ---- collectives of the lock-step multi-class fit (odx.solver.falkon_fit_lockstep) -------------------
    def _has_reduce_scatter(self):
        return dist.get_backend(self.group) == "nccl"       # RCCL; gloo has neither primitive for device-style use

    def gather_rows(self, mine, out):
        """out (world, M) <- row r = rank r's `mine` (M,).  One all-gather."""
        if not self.enabled:
            out[0].copy_(mine)
        elif self._has_reduce_scatter():
            dist.all_gather_into_tensor(out, mine, group=self.group)
        else:
            out.zero_()
            out[self.rank].copy_(mine)
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out

    def gather_blocks(self, mine, out):
        """out (world, r, D) <- block k = rank k's `mine` (r, D): one all-gather of equal-sized blocks (the centre rows each
        rank owns, padded to the largest count: odx.job.LockstepClassJob.gather_centres)."""
        if not self.enabled:
            out[0].copy_(mine)
        elif self._has_reduce_scatter():
            dist.all_gather_into_tensor(out, mine, group=self.group)
        else:
            parts = [torch.empty_like(mine) for _ in range(self.world)]       # gloo: the list form
            dist.all_gather(parts, mine, group=self.group)
            for k, p in enumerate(parts):
                out[k].copy_(p)
        return out

    def reduce_scatter_rows(self, partials, out):
        """out (M,) <- sum over ranks of their partials[self.rank]  (partials: (world, M)).  One reduce-scatter."""
        if not self.enabled:
            out.copy_(partials[0])
        elif self._has_reduce_scatter():
            dist.reduce_scatter_tensor(out, partials, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(partials[self.rank])
        return out


class EmulatedShard(RowShard):
    """ONE rank's share of a `world`-rank job, run alone in a single process: the rank's row bounds, its place among the
    owners, the collectives replaced by local copies of the right size (their calls and bytes are counted).  What comes out is
    the rank's COMPUTE — every kernel it would launch on its 1 / world of the rows, the preconditioner chains of the classes
    it owns — not the job's numbers: partial sums of the other ranks are missing, directions of the classes other ranks own
    are stand-ins (this rank's own).  For measuring the per-rank step time of a multi-GPU job on a one-GPU box
    (bench.py --emulate-world); the row indices handed to the job must lie in this rank's range (the caller folds them)."""

    def __init__(self, world, rank):
        self.group = None
        self.enabled = False
        self.world, self.rank = int(world), int(rank)
        if not 0 <= self.rank < self.world:
            raise ValueError("EmulatedShard: rank %d outside a world of %d" % (self.rank, self.world))
        self.emulated = True
        # [calls, bytes]; "centre_gather": the all-gather of the Nystroem centres' rows, one per class ((M, D) f32 received)
        self.calls = {"all_reduce": [0, 0], "all_gather": [0, 0], "reduce_scatter": [0, 0], "broadcast": [0, 0], "centre_gather": [0, 0]}

    def _count(self, kind, t):
        self.calls[kind][0] += 1
        self.calls[kind][1] += int(t.numel()) * int(t.element_size())

    def allreduce(self, v):
        self._count("all_reduce", v)
        return v

    def total(self, n_local):
        return int(n_local) * self.world

    def broadcast(self, t, src=0):
        self._count("broadcast", t)
        return t

    def gather_rows(self, mine, out):
        self._count("all_gather", out)
        out.copy_(mine.unsqueeze(0).expand_as(out))       # every owner's direction stands in as this rank's
        return out

    def gather_blocks(self, mine, out):
        self._count("centre_gather", out)
        out.copy_(mine.unsqueeze(0).expand_as(out))
        return out

    def reduce_scatter_rows(self, partials, out):
        self._count("reduce_scatter", partials)
        out.copy_(partials[self.rank])                    # this rank's partial of its own problem only
        return out
