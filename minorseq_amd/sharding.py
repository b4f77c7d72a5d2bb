"""Read sharding across the GPUs of a node and the one exchange step of the path.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).
Reads are independent and the count matrices are additive integers, so the only data-path collective is one
sum all-reduce of the [W][70] u32 count buffer (int32 view: two's-complement sums are bit-identical; the global
read count must stay below 2^31 as in the reference's int counters), preceded by a 2-int max all-reduce that
turns per-shard window bounds (min ReferenceStart, max ReferenceEnd; src/MSA.cpp:138-142) into the global window.
"""
import torch
import torch.distributed as dist

# Rehearsal switch (bench.py --force-dist): issue the collectives even in a one-rank group, so that a one-GPU box
# exercises the RCCL calls, dtypes and stream ordering of the N > 1 step.
FORCE_COLLECTIVES = False


def _collective():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def shard_range(n_reads, rank, world):
    """Contiguous read range [first, first + count) of `rank`: sizes differ by at most one read."""
    base, extra = divmod(int(n_reads), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def allreduce_bounds(begin0, end0, device):
    """Global (min begin0, max end0) over the ranks; a rank without reads passes (INT32_MAX, 0)."""
    if not _collective():
        return int(begin0), int(end0)
    t = torch.tensor([-int(begin0), int(end0)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lo, hi = t.tolist()  # one device-to-host copy
    return -int(lo), int(hi)


def allreduce_bounds_device(bounds):
    """In-place global window of a device-resident {-begin0, end0} int32 pair (msq_window_bounds_device): one
    element-wise MAX over the ranks, no host round trip."""
    if _collective():
        assert bounds.dtype == torch.int32 and bounds.numel() == 2
        dist.all_reduce(bounds, op=dist.ReduceOp.MAX)
    return bounds


def allreduce_counts(counts):
    """In-place sum of the int32 count buffer over the ranks (the RCCL all-reduce of the north star)."""
    if _collective():
        assert counts.dtype == torch.int32 and counts.is_contiguous()
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts
