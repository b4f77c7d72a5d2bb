"""msq_phase_cluster: the exact-tuple collapse of AminoAcidCaller::PhaseVariants (src/AminoAcidCaller.cpp:162-196) on
the device, against the reference's own procedure restated here -- walk the reads in order, join the first
observation with an equal tuple or open a new one: the same partition, observations in first-occurrence order."""
import numpy as np
import pytest
import torch

import cases
from minorseq_amd import abi, device, synth

pytestmark = pytest.mark.gpu


def host_partition(rows):
    seen, cluster_of, first, count = {}, [], [], []
    for r, row in enumerate(rows):
        k = row.tobytes()
        j = seen.get(k)
        if j is None:
            j = seen[k] = len(first)
            first.append(r)
            count.append(0)
        count[j] += 1
        cluster_of.append(j)
    return np.array(cluster_of), np.array(first), np.array(count)


def check(ctx, rows):
    n, K = rows.shape
    keys = torch.from_numpy(np.ascontiguousarray(rows).view(np.int16).reshape(-1)).cuda() if rows.size else torch.zeros(1, dtype=torch.int16, device="cuda")
    cl, first, count, rep = device.phase_cluster(ctx, keys, n, K)
    w_cl, w_first, w_count = host_partition(rows)
    assert (cl == w_cl).all() and (first == w_first).all() and (count == w_count).all()
    assert (rep == rows[w_first]).all()
    return len(first)


@pytest.mark.parametrize("n,K,alphabet", [(1, 1, 2), (5, 3, 1), (64, 1, 3), (65, 7, 2), (1000, 20, 2), (5000, 100, 2), (3000, 300, 511), (4097, 2, 4)])
def test_random_rows(ctx, n, K, alphabet):
    rng = np.random.RandomState(n + K)
    rows = rng.randint(0, alphabet, size=(n, K)).astype(np.uint16)
    if n >= 1000:  # make clusters of every size: most rows are copies of a few
        base = rows[rng.randint(0, 12, size=n)]
        noisy = rng.rand(n) < 0.1
        rows = np.where(noisy[:, None], rows, base).astype(np.uint16)
    c = check(ctx, rows)
    assert 1 <= c <= n


def test_rows_that_differ_in_one_late_key(ctx):
    rows = np.zeros((300, 257), dtype=np.uint16)
    rows[100:, 256] = 1      # differs only in the last key
    rows[200:, 0] = 0x1ff    # ... and only in the first
    assert check(ctx, rows) == 3
    rows = np.zeros((10, 0), dtype=np.uint16)  # no positions: every read has the same (empty) tuple
    cl, first, count, _ = device.phase_cluster(ctx, torch.zeros(1, dtype=torch.int16, device="cuda"), 10, 0, with_rows=False)
    assert (cl == 0).all() and first.tolist() == [0] and count.tolist() == [10]


def test_real_key_rows_at_bench_size(ctx):
    """Keys of msq_phase_codons on 1e6 pol reads at 20 positions (the planted clones + sequencing errors)."""
    wl = synth.workload("pol_1m")
    n = wl.n_reads
    dr = device.synth_device(ctx, wl, 0, n)
    rng = np.random.RandomState(3)
    sites = sorted({i - 1 - wl.win_begin0 for _, i, _ in wl.minor_sites})
    pos = np.array(sites[:20], dtype=np.int32)
    K = len(pos)
    keys = torch.zeros(n * K, dtype=torch.int16, device="cuda")
    abi.check(abi.lib().msq_phase_codons(ctx.h, dr.view, abi.QvThresholds.make(), wl.win_begin0, wl.W, K, abi.ptr(pos), keys.data_ptr()))
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    cl, first, count, rep = device.phase_cluster(ctx, keys, n, K)
    dt = time.perf_counter() - t0
    rows = keys.cpu().numpy().view(np.uint16).reshape(n, K)
    w_cl, w_first, w_count = host_partition(rows)
    assert (cl == w_cl).all() and (first == w_first).all() and (count == w_count).all() and (rep == rows[w_first]).all()
    assert count[0] > 0.6 * n and len(first) > 100  # the major clone (60 columns free of errors), the two minors, the error tuples
    print("msq_phase_cluster: %d reads x %d keys -> %d clusters, %.2f ms incl. result copies" % (n, K, len(first), dt * 1e3))
