#!/usr/bin/env python3
"""One-off soak of the paths the small fuzz never reaches (launches of many batches per wave, merged slab launches,
rounds, the carrier rule, the lookup gather): random synthetic samples of 10^4 .. 10^5 reads, stream kernel against
the simple kernel bit for bit, the lookup gather against the streaming one.  usage: tools/soak_big.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minorseq_amd import abi, device, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = device.Ctx(0)
L = abi.lib()
NONE = abi.QvThresholds.make()
t0 = time.time()
n_cases = 0
while time.time() - t0 < budget:
    name = str(rng.choice(["pol_100k", "hiv_mix_200k", "amp20k_10m"]))
    n = int(rng.choice([9000, 20000, 45000, 90000])) if name != "amp20k_10m" else int(rng.choice([6000, 24000, 50000]))
    kw = dict(seed=int(rng.randint(1, 1 << 30)), sub=float(rng.choice([0.0, 0.0005, 0.004, 0.02])), ins=float(rng.choice([0.0, 0.001, 0.02])),
              dele=float(rng.choice([0.0, 0.004, 0.03])), n_minors=int(rng.choice([0, 1, 2, 4])), minor_frac=float(rng.choice([0.01, 0.1, 0.3])),
              minor_codons=int(rng.choice([5, 45, 300])))
    if rng.rand() < 0.4:
        kw.update(jitter_frac=float(rng.choice([0.2, 0.9])), max_jitter=int(rng.choice([100, 2500])))
    wl = synth.workload(name, n_reads=n, **kw)
    dr = device.synth_device(ctx, wl, 0, n)
    b, W = wl.win_begin0, wl.W
    if rng.rand() < 0.3:  # empty columns around the reads
        pad = int(rng.randint(1, 900))
        b, W = max(0, b - pad), W + 2 * pad
    want = device.pileup(ctx, dr, NONE, b, W, mode=abi.MSQ_PILEUP_SIMPLE)
    got = device.pileup(ctx, dr, NONE, b, W)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), ("pileup", name, n, kw, b, W)
    if W > 4096 and rng.rand() < 0.5:
        part = device.Partition(ctx, dr, b, W)
        got = part.pileup(NONE)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), ("partition", name, n, kw, b, W)
        part.close()
    K = int(rng.choice([3, 20, 70]))
    pos = np.sort(rng.randint(-2, W + 2, size=K)).astype(np.int32)
    keys = []
    for mode in ("1", "0"):
        os.environ["MSQ_PHASE_LOOKUP"] = mode
        k = torch.zeros(n * K, dtype=torch.int16, device="cuda")
        abi.check(L.msq_phase_codons(ctx.h, dr.view, NONE, b, W, K, abi.ptr(pos), k.data_ptr()))
        keys.append(k)
    del os.environ["MSQ_PHASE_LOOKUP"]
    assert torch.equal(keys[0], keys[1]), ("phase", name, n, kw, b, W, pos.tolist())
    abi.check(L.msq_pileup_check(ctx.h))
    n_cases += 1
    if n_cases % 10 == 0:
        print("%d cases, %.0f s" % (n_cases, time.time() - t0), flush=True)
print("soak ok: %d cases in %.0f s (merged launches %d, lookups %d)" % (n_cases, time.time() - t0, L.msq_debug_merged_launches(), L.msq_debug_phase_lookups()))
