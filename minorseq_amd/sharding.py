"""Read sharding across the GPUs of a node and the one exchange step of the path.

One process per GPU.  Reads are independent and the count matrices are additive integers, so the only data-path
exchange is one u32 sum of the [W][70] count buffer, preceded by (or grouped with) a 2-int MAX that turns per-shard
window bounds (min ReferenceStart, max ReferenceEnd; src/MSA.cpp:138-142) into the global window.

On the GPU the exchange is the library's own: `LibExchange` drives msq_comm_* (include/msq_abi.h), RCCL over xGMI on
the context's stream; torch.distributed is only the control plane there (handing the communicator id around, the
bench's barrier and max-over-ranks clock).  `TorchExchange` is the same interface over torch.distributed tensors --
the CPU rehearsal of the arithmetic (gloo, tests/test_sharding_gloo.py).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import abi


def shard_range(n_reads, rank, world):
    """Contiguous read range [first, first + count) of `rank`: [rank * N / world, (rank + 1) * N / world), the rule of
    msq_comm_shard; sizes differ by at most one read."""
    n, rank, world = int(n_reads), int(rank), int(world)
    lo, hi = n * rank // world, n * (rank + 1) // world
    return lo, hi - lo


def shard_range_bases(prefix, rank, world):
    """The same, balanced by base count (msq_comm_shard_bases): prefix[i] = bases (or groups) of the reads before read
    i, n + 1 entries; rank r starts at the first read i with prefix[i] >= ceil(r * prefix[n] / world)."""
    n = len(prefix) - 1
    if n <= 0:
        return 0, 0
    total = int(prefix[n])

    def boundary(r):
        if r <= 0:
            return 0
        if r >= world:
            return n
        want = (total * r + world - 1) // world
        return int(np.searchsorted(np.asarray(prefix), want, side="left"))
    lo, hi = boundary(rank), boundary(rank + 1)
    return lo, max(0, hi - lo)


class LibExchange:
    """The library's RCCL communicator bound to a device.Ctx (msq_comm_create).  `bootstrap(id_bytes_or_None)` must
    return rank 0's 128-byte id on every rank (rank 0 passes its own in, the others None)."""

    def __init__(self, ctx, rank, world, bootstrap):
        self.ctx, self.rank, self.world = ctx, rank, world
        L = abi.lib()
        uid = (C.c_uint8 * abi.MSQ_COMM_ID_BYTES)()
        if rank == 0:
            abi.check(L.msq_comm_unique_id(uid))
        raw = bootstrap(bytes(uid) if rank == 0 else None)
        uid = (C.c_uint8 * abi.MSQ_COMM_ID_BYTES).from_buffer_copy(raw)
        abi.check(L.msq_comm_create(ctx.h, rank, world, uid))
        assert (L.msq_comm_rank(ctx.h), L.msq_comm_world(ctx.h)) == (rank, world)

    def allreduce_bounds_device(self, bounds):
        assert bounds.dtype == torch.int32 and bounds.numel() == 2
        abi.check(abi.lib().msq_comm_allreduce_bounds(self.ctx.h, bounds.data_ptr()))
        return bounds

    def allreduce_counts(self, counts):
        assert counts.dtype == torch.int32 and counts.is_contiguous()
        abi.check(abi.lib().msq_comm_allreduce_counts(self.ctx.h, counts.data_ptr(), counts.numel()))
        return counts

    def allreduce_step(self, bounds, counts):
        """bounds MAX + counts SUM as one RCCL group (one launch)."""
        assert bounds.dtype == torch.int32 and bounds.numel() == 2 and counts.dtype == torch.int32 and counts.is_contiguous()
        abi.check(abi.lib().msq_comm_allreduce_step(self.ctx.h, bounds.data_ptr(), counts.data_ptr(), counts.numel()))

    def allreduce_bounds(self, begin0, end0):
        t = torch.tensor([-int(begin0), int(end0)], dtype=torch.int32, device=torch.device("cuda", self.ctx.device))
        lo, hi = self.allreduce_bounds_device(t).tolist()
        return -int(lo), int(hi)

    def close(self):
        abi.check(abi.lib().msq_comm_destroy(self.ctx.h))


class TorchExchange:
    """The same exchange over torch.distributed tensors (CPU rehearsal with gloo; a no-op outside a process group).
    group: the process group to use (default: the world group)."""

    def __init__(self, group=None):
        self.group = group

    @staticmethod
    def _on():
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def allreduce_bounds_device(self, bounds):
        if self._on():
            assert bounds.dtype == torch.int32 and bounds.numel() == 2
            dist.all_reduce(bounds, op=dist.ReduceOp.MAX, group=self.group)
        return bounds

    def allreduce_counts(self, counts):
        if self._on():
            assert counts.dtype == torch.int32 and counts.is_contiguous()
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)  # int32 view: two's-complement sums are the u32 sums
        return counts

    def allreduce_step(self, bounds, counts):
        self.allreduce_bounds_device(bounds)
        self.allreduce_counts(counts)

    def allreduce_bounds(self, begin0, end0, device="cpu"):
        if not self._on():
            return int(begin0), int(end0)
        t = torch.tensor([-int(begin0), int(end0)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        lo, hi = t.tolist()
        return -int(lo), int(hi)

    def close(self):
        pass


class NoExchange(TorchExchange):
    """One GPU, no process group: the shard is the sample."""

    @staticmethod
    def _on():
        return False


def torch_bootstrap(raw):
    """Hand rank 0's communicator id to every rank through the (control-plane) torch process group."""
    box = [raw]
    dist.broadcast_object_list(box, src=0)
    return box[0]


# module-level forms of the torch exchange (tests/test_sharding_gloo.py)
_TORCH = TorchExchange()


def allreduce_bounds(begin0, end0, device):
    return _TORCH.allreduce_bounds(begin0, end0, device)


def allreduce_bounds_device(bounds):
    return _TORCH.allreduce_bounds_device(bounds)


def allreduce_counts(counts):
    return _TORCH.allreduce_counts(counts)
