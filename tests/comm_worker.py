"""One rank of tests/test_gpu_comm.py's multi-process case: the library communicator's entry points with world > 1 on
one GPU (MSQ_COMM_TRANSPORT=shm).  argv: rank world id_hex.  Prints OK or raises."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minorseq_amd import abi, device, sharding  # noqa: E402


def main():
    rank, world, id_hex = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    raw = bytes.fromhex(id_hex).ljust(abi.MSQ_COMM_ID_BYTES, b"\0")
    ctx = device.Ctx(0)
    ex = sharding.LibExchange(ctx, rank, world, lambda _: raw)
    L = abi.lib()
    dev = torch.device("cuda", 0)
    # window bounds: {-begin, end} MAX
    assert ex.allreduce_bounds(100 + 7 * rank, 5000 - 3 * rank) == (100, 5000)
    # counts: u32 SUM, a buffer of one slot and one of several (12 MB per rank: the transport moves 8 MB at a time)
    for n in (2843 * 70, 3_000_001):
        mine = [np.random.default_rng(1000 * r + 1).integers(0, 2**31, n, dtype=np.int64).astype(np.uint32) for r in range(world)]
        want = np.zeros(n, dtype=np.uint32)
        for m in mine:
            want += m
        t = torch.from_numpy(mine[rank].view(np.int32).copy()).to(dev)
        ex.allreduce_counts(t)
        assert (t.cpu().numpy().view(np.uint32) == want).all(), n
    # the step's group: both at once
    b = torch.tensor([-(50 + rank), 900 + rank], dtype=torch.int32, device=dev)
    c = torch.full((1000,), rank + 1, dtype=torch.int32, device=dev)
    ex.allreduce_step(b, c)
    assert b.tolist() == [-50, 900 + world - 1] and (c.cpu().numpy() == world * (world + 1) // 2).all()
    # all-gather: every rank's bytes in rank order (again more than a slot)
    for nbytes in (6, 9_000_000):
        send = torch.full((nbytes,), 17 * rank + 1, dtype=torch.uint8, device=dev)
        recv = torch.zeros(nbytes * world, dtype=torch.uint8, device=dev)
        abi.check(L.msq_comm_allgather(ctx.h, send.data_ptr(), recv.data_ptr(), nbytes))
        ctx.sync()
        got = recv.cpu().numpy().reshape(world, nbytes)
        assert all((got[r] == 17 * r + 1).all() for r in range(world))
    # one status for all ranks: the lowest failing rank's code and text, on every rank
    agreed = C.c_int(0)
    abi.check(L.msq_comm_agree_status(ctx.h, abi.MSQ_OK, None, C.byref(agreed)))
    assert agreed.value == abi.MSQ_OK
    mine_rc = abi.MSQ_ERR_CIGAR if rank >= 1 else abi.MSQ_OK
    text = ("Unexpected cigar %d" % (70 + rank)).encode() if rank >= 1 else None
    abi.check(L.msq_comm_agree_status(ctx.h, mine_rc, text, C.byref(agreed)))
    assert agreed.value == abi.MSQ_ERR_CIGAR and L.msq_last_error() == b"Unexpected cigar 71"
    ex.close()
    print("OK rank %d of %d" % (rank, world))


if __name__ == "__main__":
    main()
