"""The device packer (msq_pack_reads_device: ArrayBase records -> packed reads on the GPU) against the host packer
(msq_pack_reads), which is the format's specification: identical tables and bit planes, identical QV bytes wherever
a slot holds a base, the same error for an unknown cigar char; one chunk and many chunks, one thread and many; QVs
present on some bases only (absent -> 255, planes appearing late)."""
import numpy as np
import pytest

import cases
from minorseq_amd import abi, device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return device.Ctx(0)


def assert_same_packing(dev, host, base_off):
    assert dev["n_reads"] == host["n_reads"] and dev["n_groups"] == host["n_groups"] and dev["flags"] == host["flags"]
    for f in ("ref_start", "ref_end", "group_off", "bop"):
        assert (dev[f] == host[f]).all(), f
    g = host["n_groups"]
    if g == 0:
        return
    # which QV bytes hold a base: slots [0, up) and (31 - down, 31] of every group
    lens = np.diff(base_off)
    goff = host["group_off"].astype(np.int64)
    n_in = np.zeros(g, dtype=np.int64)
    for r, L in enumerate(lens):
        k = goff[r + 1] - goff[r]
        if k:
            n_in[goff[r]:goff[r + 1]] = np.minimum(32, L - 32 * np.arange(k))
    o1 = host["bop"].reshape(g, 4)[:, 3]
    slot = np.arange(32)
    cons_mask = ((o1[:, None] >> slot) & 1) == 0  # ops 0, 1
    up = cons_mask.sum(axis=1)
    held = cons_mask | (slot[None, :] >= 32 - (n_in - up)[:, None])
    byte_of = ((slot & 7) << 2) | (slot >> 3)  # MSQ_QV_BYTE
    held_bytes = np.zeros((g, 32), dtype=bool)
    held_bytes[:, byte_of] = held
    for f in ("qual", "delqv", "subqv", "insqv"):
        assert (dev[f] is None) == (host[f] is None), f
        if host[f] is not None:
            d, h = dev[f].reshape(g, 32), host[f].reshape(g, 32)
            assert (d[held_bytes] == h[held_bytes]).all(), f


@pytest.mark.parametrize("name", sorted(cases.edge_sets()))
def test_edge_sets(ctx, name):
    reads, _ = cases.edge_sets()[name]
    host = device.pack_host(reads)
    dev = device.pack_device(ctx, reads)
    assert_same_packing(dev, host, reads["base_off"])
    for f in ("qual", "delqv", "subqv", "insqv"):  # one chunk: even the unused bytes agree (both zero)
        if host[f] is not None:
            assert (dev[f] == host[f]).all()


@pytest.mark.parametrize("rich", [False, True])
@pytest.mark.parametrize("chunk,threads", [(0, 0), (1, 1), (700, 3), (5000, 16), (1 << 18, 8)])
def test_random_reads_chunks_and_threads(ctx, rich, chunk, threads):
    reads = cases.random_reads(17 + rich, 400, 333, rich=rich, exotic=True)
    host = device.pack_host(reads)
    dev = device.pack_device(ctx, reads, threads=threads, chunk_bases=chunk)
    assert_same_packing(dev, host, reads["base_off"])


def test_no_qvs_and_empty(ctx):
    reads = cases.random_reads(3, 50, 100)
    reads["qual"] = None
    host = device.pack_host(reads)
    dev = device.pack_device(ctx, reads)
    assert dev["qual"] is None and dev["subqv"] is None
    assert_same_packing(dev, host, reads["base_off"])
    empty = cases.make_reads([])
    dev = device.pack_device(ctx, empty)
    assert dev["n_reads"] == 0 and dev["n_groups"] == 0
    zero_len = cases.make_reads([(5, "", ""), (5, "===", "ACG"), (9, "", "")])
    assert_same_packing(device.pack_device(ctx, zero_len), device.pack_host(zero_len), zero_len["base_off"])


@pytest.mark.parametrize("chunk", [0, 900])
def test_partial_presence(ctx, chunk):
    """QVs on some bases only; with small chunks the planes appear after the first chunks were packed."""
    reads = cases.random_reads(29, 300, 200, rich=True)
    total = int(reads["base_off"][-1])
    rng = np.random.RandomState(1)
    hq = (rng.rand(total) < 0.5).astype(np.uint8)
    hr = (rng.rand(total) < 0.3).astype(np.uint8)
    first = int(reads["base_off"][120])  # nothing present in the first 120 reads
    hq[:first] = 0
    hr[:first + 5000] = 0
    want = dict(reads)
    want["qual"] = np.where(hq, reads["qual"], 255).astype(np.uint8)
    for f in ("subqv", "delqv", "insqv"):
        want[f] = np.where(hr, reads[f], 255).astype(np.uint8)
    host = device.pack_host(want)
    dev = device.pack_device(ctx, reads, has_qual=hq, has_rich=hr, chunk_bases=chunk, threads=2)
    assert_same_packing(dev, host, reads["base_off"])


def test_unexpected_cigar(ctx):
    reads = cases.random_reads(5, 200, 150)
    off = reads["base_off"]
    reads["cigar"][int(off[150]) + 3] = ord("N")  # later in read order
    reads["cigar"][int(off[40]) + 7] = ord("M")   # the first one: the one the reference throws on (src/MSA.cpp:182)
    for chunk in (0, 2000):
        with pytest.raises(abi.MsqError) as e:
            device.pack_device(ctx, reads, chunk_bases=chunk)
        assert e.value.code == abi.MSQ_ERR_CIGAR and "Unexpected cigar %d" % ord("M") in str(e.value)
        for threads in (1, 5):  # the packer that applies thresholds translates on the host (compact staging): the same report
            with pytest.raises(abi.MsqError) as eb:
                device.pack_device(ctx, reads, chunk_bases=chunk, threads=threads, thr=abi.QvThresholds.make(qual=20))
            assert str(eb.value) == str(e.value)
    with pytest.raises(abi.MsqError) as e2:
        device.pack_host(reads)
    assert str(e2.value) == str(e.value)


def test_pileup_on_device_packed_reads(ctx):
    """Counts from device-packed reads equal the counts from host-packed reads (and so the oracle's)."""
    reads = cases.random_reads(41, 600, 500, rich=True)
    thr = abi.QvThresholds.make(qual=30, del_=20, sub=42, ins=10)
    out = []
    for packed in (device.pack_host(reads), device.pack_device(ctx, reads, chunk_bases=40000)):
        dr = device.upload(packed)
        b0, e0 = device.window_bounds(ctx, dr)
        nuc, codon = device.pileup(ctx, dr, thr, b0, e0 - b0)
        out.append((b0, e0, nuc.cpu().numpy().copy(), codon.cpu().numpy().copy()))
    assert out[0][:2] == out[1][:2]
    assert (out[0][2] == out[1][2]).all() and (out[0][3] == out[1][3]).all()
