"""The N > 1 path on CPU: two gloo ranks shard a read set, pile their shards up (with the oracle standing in for
the GPU kernel, which needs a GPU), combine window bounds and count matrices with minorseq_amd.sharding -- the
same functions bench.py runs over RCCL -- and must reproduce the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from minorseq_amd import sharding, synth
from oracle import orc


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            parts = [sharding.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_shard_range_is_the_library_rule():
    """msq_comm_shard (host arithmetic, no GPU and no RCCL needed): [r N / P, (r + 1) N / P)."""
    import ctypes as C
    from minorseq_amd import abi
    for n in (0, 5, 1000003, 2 ** 40 + 5):
        for world in (1, 2, 3, 8):
            for r in range(world):
                f, c = C.c_uint64(), C.c_uint64()
                abi.check(abi.lib().msq_comm_shard(n, r, world, C.byref(f), C.byref(c)))
                assert (f.value, c.value) == sharding.shard_range(n, r, world)


def test_shard_by_base_count_is_the_library_rule_and_balances():
    """msq_comm_shard_bases (host arithmetic): contiguous ranges that tile the reads, equal to the Python rule, and --
    on hiv_1k's staggered reads, a tenth of them shortened -- closer to equal BASE counts than equal read counts are
    ("balanced by base count via base_offset")."""
    import ctypes as C
    from minorseq_amd import abi
    wl = synth.workload("hiv_1k", n_reads=1000)
    reads = wl.host_reads(0, 1000)
    sizes = np.diff(reads["base_off"]).astype(np.uint64)
    # the worst case for equal read counts: the long reads first, short ones behind
    for order in (np.arange(1000), np.argsort(-sizes.astype(np.int64), kind="stable")):
        sz = sizes[order]
        prefix = np.concatenate([[0], np.cumsum(sz)]).astype(np.uint64)
        for world in (1, 2, 3, 8):
            parts = []
            for r in range(world):
                f, c = C.c_uint64(), C.c_uint64()
                abi.check(abi.lib().msq_comm_shard_bases(prefix.ctypes.data, 1000, r, world, C.byref(f), C.byref(c)))
                assert (f.value, c.value) == sharding.shard_range_bases(prefix, r, world)
                parts.append((f.value, c.value))
            assert parts[0][0] == 0 and sum(c for _, c in parts) == 1000
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            by_bases = [int(prefix[f + c] - prefix[f]) for f, c in parts]
            by_reads = [int(prefix[f + c] - prefix[f]) for f, c in (sharding.shard_range(1000, r, world) for r in range(world))]
            assert max(by_bases) <= max(by_reads) + int(sz.max())
            assert max(by_bases) - min(by_bases) <= 2 * int(sz.max())
    # equal reads: the read-count rule up to one read
    prefix = (np.arange(1001) * 2850).astype(np.uint64)
    for world in (2, 3, 8):
        for r in range(world):
            f, c = sharding.shard_range_bases(prefix, r, world)
            f0, c0 = sharding.shard_range(1000, r, world)
            assert abs(f - f0) <= 1 and abs(c - c0) <= 1
    # no reads, and more ranks than reads
    f, c = C.c_uint64(7), C.c_uint64(7)
    abi.check(abi.lib().msq_comm_shard_bases(None, 0, 0, 4, C.byref(f), C.byref(c)))
    assert (f.value, c.value) == (0, 0)
    prefix = np.array([0, 5, 9], dtype=np.uint64)
    got = [sharding.shard_range_bases(prefix, r, 4) for r in range(4)]
    assert sum(c for _, c in got) == 2 and all(c in (0, 1) for _, c in got)


def test_bench_launcher_dry_run():
    """`python bench.py --gpus N` (N > 1) without a launcher around it starts its own rank processes: the dry run prints
    every child's command and environment -- one fresh interpreter per GPU, loopback rendezvous on a free port, the
    same arguments -- and touches no GPU."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2", "--launch-dry-run"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    ranks = plan["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1, 2, 3]
    ports = {r["env"]["MASTER_PORT"] for r in ranks}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for k, r in enumerate(ranks):
        e = r["env"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"]) == (str(k), str(k), "4", "127.0.0.1")
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert r["argv"][0] == sys.executable
        assert "--launch-dry-run" not in r["argv"] and r["argv"][-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    # under a launcher's WORLD_SIZE the flag is an error, not a silent run
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--launch-dry-run", "--cpu-sample", "0"],
                         capture_output=True, text=True, timeout=120, env=env2)
    assert bad.returncode != 0 and "launch-dry-run" in bad.stderr


def _rank_main(rank, world, port, n_reads, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = synth.workload("hiv_1k", n_reads=n_reads)  # staggered reads: shards see different windows
        first, count = sharding.shard_range(n_reads, rank, world)
        reads = wl.host_reads(first, count)
        b0 = int(reads["ref_start"].min()) if count else 2 ** 31 - 1
        e0 = int(reads["ref_end"].max()) if count else 0
        gb0, ge0 = sharding.allreduce_bounds(b0, e0, "cpu")
        # the device-resident form of the same merge ({-begin0, end0}, one MAX; what msq_window_bounds_device feeds)
        pair = torch.tensor([-b0, e0], dtype=torch.int32)
        sharding.allreduce_bounds_device(pair)
        assert pair.tolist() == [-gb0, ge0]
        W = ge0 - gb0
        counts = torch.zeros(W * 70, dtype=torch.int32)
        if count:
            # The GPU path piles every shard up inside the GLOBAL window (msq_pileup takes win_begin0 and W).  The
            # oracle derives its window from its reads, so give the shard one extra read that spans the global
            # window and consists of a single 'P' op: it moves the bounds (src/MSA.cpp:138-142) and adds nothing.
            reads = dict(reads)
            reads["n_reads"] = count + 1
            reads["ref_start"] = np.append(reads["ref_start"], np.int32(gb0))
            reads["ref_end"] = np.append(reads["ref_end"], np.int32(ge0))
            reads["base_off"] = np.append(reads["base_off"], reads["base_off"][-1] + 1)
            reads["cigar"] = np.append(reads["cigar"], np.uint8(ord("P")))
            reads["nuc"] = np.append(reads["nuc"], np.uint8(ord("*")))
            reads["qual"] = np.append(reads["qual"], np.uint8(0))
            m = orc.Msa(reads)
            assert (m.begin1 - 1, m.W) == (gb0, W)
            counts[:W * 6] = torch.from_numpy(m.nuc_counts()).reshape(-1)
            counts[W * 6:] = torch.from_numpy(m.codon_counts()).reshape(-1)
        sharding.allreduce_counts(counts)
        if rank == 0:
            np.save(out_path, np.concatenate([[gb0, ge0], counts.numpy()]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_reads", [61, 200])
def test_two_ranks_equal_one(tmp_path, n_reads):
    out = str(tmp_path / "r0.npy")
    mp.spawn(_rank_main, args=(2, _free_port(), n_reads, out), nprocs=2, join=True)
    got = np.load(out)
    wl = synth.workload("hiv_1k", n_reads=n_reads)
    m = orc.Msa(wl.host_reads(0, n_reads))
    assert (int(got[0]) + 1, int(got[1]) + 1) == (m.begin1, m.end1)
    W = m.W
    assert (got[2:2 + W * 6].reshape(W, 6) == m.nuc_counts()).all()
    assert (got[2 + W * 6:].reshape(W, 64) == m.codon_counts()).all()
