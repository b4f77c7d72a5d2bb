#!/usr/bin/env python3
"""Time the phasing gather (msq_phase_codons) on a synthetic workload resident in HBM:
python tools/bench_phase.py [--workload pol_1m] [--reads N] [--positions K] [--steps S] -> one JSON line."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from minorseq_amd import abi, device, synth

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pol_1m")
ap.add_argument("--reads", type=int, default=0)
ap.add_argument("--positions", type=int, default=20)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
wl = synth.workload(a.workload)
n = a.reads or wl.n_reads
ctx = device.Ctx(0)
dr = device.synth_device(ctx, wl, 0, n)
b0, e0 = device.window_bounds(ctx, dr)
W = e0 - b0
pos = np.linspace(1, W - 4, a.positions).astype(np.int32)  # K codon starts spread over the window
keys = torch.zeros(n * len(pos), dtype=torch.int16, device="cuda")
thr = abi.QvThresholds.make()
L = abi.lib()
def step():
    abi.check(L.msq_phase_codons(ctx.h, dr.view, thr, b0, W, len(pos), abi.ptr(pos), keys.data_ptr()))  # synchronises
step()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
dt = (time.perf_counter() - t0) / a.steps
# the collapse of equal key rows (msq_phase_cluster; synchronises), outputs allocated once
import ctypes as C
K = len(pos)
outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(3)]
rep = torch.zeros(n * K, dtype=torch.int16, device="cuda")
nc = C.c_uint64(0)
def cluster():
    abi.check(L.msq_phase_cluster(ctx.h, keys.data_ptr(), n, K, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), rep.data_ptr(), n, C.byref(nc)))
cluster()
t1 = time.perf_counter()
for _ in range(a.steps):
    cluster()
dt_cluster = (time.perf_counter() - t1) / a.steps
alg = wl.algorithmic_bytes(n_reads=n)
print(json.dumps({"kernel": "pileup_kernel<NP,1,PHASE>", "workload": wl.name, "reads": n, "positions": len(pos), "ms": dt * 1e3,
                  "reads_per_s": n / dt, "algorithmic_bytes": alg, "achieved_GBs": alg / dt / 1e9, "frac_of_8TBs": alg / dt / 8e12,
                  "cluster_ms": dt_cluster * 1e3, "clusters": int(nc.value)}))
