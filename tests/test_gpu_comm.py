"""The exchange step behind the C ABI (msq_comm_*, RCCL on the context's stream) with the one rank a one-GPU box
offers: the communicator comes up from its 128-byte id, every collective leaves a one-rank buffer as it was, the
whole sharded step (prologue -> pileup -> one RCCL group -> testing pass) equals the plain calls, and the shard
rule partitions the reads.  The N = 2 arithmetic is covered on CPU (tests/test_sharding_gloo.py)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import cases
from minorseq_amd import abi, device, sharding, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cx():
    c = device.Ctx(0)
    ex = sharding.LibExchange(c, 0, 1, lambda raw: raw)
    yield c, ex
    ex.close()
    c.close()


def test_pileup_begin_bounds_and_cleared_matrices(ctx):
    L = abi.lib()
    for reads in (cases.random_reads(5, 300, 700, p_ins=0.05), cases.random_reads(6, 3, 90), cases.make_reads([])):
        dr = device.upload(device.pack_host(reads), ctx.device)
        b, e = device.window_bounds(ctx, dr)
        for n_words in (0, 1, 5, 70 * 701, 4096):
            buf = torch.full((max(n_words, 1) + 8,), 7, dtype=torch.int32, device="cuda")
            bounds = torch.full((2,), 99, dtype=torch.int32, device="cuda")
            abi.check(L.msq_pileup_begin(ctx.h, dr.view, bounds.data_ptr(), buf[4:].data_ptr(), n_words))  # 16-byte aligned start
            assert bounds.tolist() == [-b, e]
            got = buf.cpu().numpy()
            assert (got[:4] == 7).all() and (got[4:4 + n_words] == 0).all() and (got[4 + n_words:] == 7).all()
            if n_words > 3:  # a start that is not 16-byte aligned
                buf.fill_(7)
                abi.check(L.msq_pileup_begin(ctx.h, dr.view, bounds.data_ptr(), buf[1:].data_ptr(), n_words - 3))
                got = buf.cpu().numpy()
                assert got[0] == 7 and (got[1:n_words - 2] == 0).all() and (got[n_words - 2:] == 7).all()


def test_one_rank_collectives_are_identities(cx):
    ctx, ex = cx
    L = abi.lib()
    assert (L.msq_comm_rank(ctx.h), L.msq_comm_world(ctx.h)) == (0, 1)
    bounds = torch.tensor([-2252, 5095], dtype=torch.int32, device="cuda")
    counts = torch.arange(70 * 300, dtype=torch.int32, device="cuda")
    want = counts.clone()
    ex.allreduce_bounds_device(bounds)
    ex.allreduce_counts(counts)
    ex.allreduce_step(bounds, counts)
    assert bounds.tolist() == [-2252, 5095] and torch.equal(counts, want)
    assert ex.allreduce_bounds(2252, 5095) == (2252, 5095)
    out = torch.zeros_like(counts)
    abi.check(L.msq_comm_allgather(ctx.h, counts.data_ptr(), out.data_ptr(), counts.numel() * 4))
    assert torch.equal(out, want)
    # a second communicator on the same context is refused, bad ranks too
    uid = (C.c_uint8 * abi.MSQ_COMM_ID_BYTES)()
    assert L.msq_comm_create(ctx.h, 0, 1, uid) == abi.MSQ_ERR_ARG
    other = device.Ctx(0)
    assert L.msq_comm_create(other.h, 2, 2, uid) == abi.MSQ_ERR_ARG
    # no communicator: every exchange is a no-op
    abi.check(L.msq_comm_allreduce_step(other.h, bounds.data_ptr(), counts.data_ptr(), counts.numel()))
    assert torch.equal(counts, want)
    other.close()


def test_status_agreement(cx):
    """msq_comm_agree_status (a rank that alone sees an error must not throw before the others know): with
    the one rank there is, the status and its text go through the all-gather and come back as they went in; without a
    communicator the call is the identity.  msq_comm_available is the local check ranks make before the collective
    msq_comm_create."""
    ctx, ex = cx
    L = abi.lib()
    assert L.msq_comm_available() == abi.MSQ_OK
    agreed = C.c_int(99)
    abi.check(L.msq_comm_agree_status(ctx.h, abi.MSQ_OK, None, C.byref(agreed)))
    assert agreed.value == abi.MSQ_OK
    abi.check(L.msq_comm_agree_status(ctx.h, abi.MSQ_ERR_CIGAR, b"Unexpected cigar 78", C.byref(agreed)))
    assert agreed.value == abi.MSQ_ERR_CIGAR and L.msq_last_error() == b"Unexpected cigar 78"
    # the text of the library's own last error when none is handed in
    assert L.msq_comm_create(ctx.h, 0, 1, None) == abi.MSQ_ERR_ARG
    text = L.msq_last_error()
    abi.check(L.msq_comm_agree_status(ctx.h, abi.MSQ_ERR_ARG, None, C.byref(agreed)))
    assert agreed.value == abi.MSQ_ERR_ARG and L.msq_last_error() == text and b"msq_comm_create" in text
    other = device.Ctx(0)
    abi.check(L.msq_comm_agree_status(other.h, abi.MSQ_ERR_BASE, b"Unexpected base", C.byref(agreed)))
    assert agreed.value == abi.MSQ_ERR_BASE and L.msq_last_error() == b"Unexpected base"
    other.close()


def test_sharded_step_equals_plain_calls(cx):
    """Two shards of one sample pushed through the sharded step one after the other on the one rank there is, added up
    by hand where the all-reduce would add them: the global window and the whole sample's matrices and calls."""
    ctx, ex = cx
    L = abi.lib()
    wl = synth.workload("hiv_1k", n_reads=900)  # staggered reads: the shards see different windows
    thr = abi.QvThresholds.make()
    whole = device.synth_device(ctx, wl, 0, 900)
    b0, e0 = device.window_bounds(ctx, whole)
    W = e0 - b0
    want_n, want_c = device.pileup(ctx, whole, thr, b0, W)
    total = torch.zeros(W * 70, dtype=torch.int32, device="cuda")
    glob = torch.tensor([-(2 ** 31 - 1), 0], dtype=torch.int32, device="cuda")
    for r in range(2):
        first, n = sharding.shard_range(900, r, 2)
        shard = device.synth_device(ctx, wl, first, n)
        counts = torch.full((W * 70,), -1, dtype=torch.int32, device="cuda")
        bounds = torch.zeros(2, dtype=torch.int32, device="cuda")
        abi.check(L.msq_pileup_begin(ctx.h, shard.view, bounds.data_ptr(), counts.data_ptr(), counts.numel()))
        abi.check(L.msq_pileup(ctx.h, shard.view, thr, b0, W, counts[:W * 6].data_ptr(), counts[W * 6:].data_ptr(), abi.MSQ_PILEUP_AUTO))
        ex.allreduce_step(bounds, counts)
        abi.check(L.msq_pileup_check(ctx.h))
        total += counts
        glob = torch.maximum(glob, bounds)
    assert glob.tolist() == [-b0, e0]
    assert torch.equal(total[:W * 6], want_n) and torch.equal(total[W * 6:], want_c)


def test_shard_rule_matches_the_library():
    L = abi.lib()
    for n in (0, 1, 7, 1000003, 2 ** 40 + 5):
        for world in (1, 2, 3, 8):
            nxt = 0
            for r in range(world):
                f, c = C.c_uint64(), C.c_uint64()
                abi.check(L.msq_comm_shard(n, r, world, C.byref(f), C.byref(c)))
                assert (f.value, c.value) == sharding.shard_range(n, r, world) and f.value == nxt
                nxt += c.value
            assert nxt == n


@pytest.mark.parametrize("world", [2, 4])
def test_entry_points_with_several_ranks_over_the_rehearsal_transport(world, tmp_path):
    """msq_comm_* with world > 1 on the one-GPU box: `world` processes, MSQ_COMM_TRANSPORT=shm (RCCL refuses ranks that share a
    GPU; the transport is the library's own, behind the same entry points) -- bounds MAX, counts SUM beyond one transfer
    slot, the step's group, all-gather in rank order, the agreed status of a run in which two ranks fail
    (tests/comm_worker.py).  What stays untested without a multi-GPU node is RCCL itself."""
    import subprocess
    import sys
    env = dict(os.environ, MSQ_COMM_TRANSPORT="shm")
    ident = os.urandom(16).hex()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "comm_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), ident], env=env, stdout=open(tmp_path / ("out%d" % r), "w"),
                              stderr=open(tmp_path / ("err%d" % r), "w")) for r in range(world)]
    for p in procs:
        p.wait(timeout=280)
    for r, p in enumerate(procs):
        assert p.returncode == 0, open(tmp_path / ("err%d" % r)).read()[-2000:]
        assert ("OK rank %d of %d" % (r, world)) in open(tmp_path / ("out%d" % r)).read()
