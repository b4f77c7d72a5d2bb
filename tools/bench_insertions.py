#!/usr/bin/env python3
"""Time the insertion scan (msq_insertion_counts / msq_insertion_records) and the pileup on plain-layout
reads: `--reads` reads of a synthetic workload, unpacked on the host, packed on the device with MSQ_PACK_PLAIN.

  python tools/bench_insertions.py [--workload pol_100k] [--reads 100000] [--out file.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pol_100k")
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    from minorseq_amd import abi, device, synth

    ctx = device.Ctx(0)
    ctx.use_current_stream()
    wl = synth.workload(a.workload)
    reads = wl.host_reads(0, a.reads)
    t0 = time.perf_counter()
    packed = device.pack_device(ctx, reads, plain=True)
    t_pack = time.perf_counter() - t0
    dr = device.upload(packed)
    b0, e0 = device.window_bounds(ctx, dr)
    W = e0 - b0
    thr = abi.QvThresholds.make()

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / reps

    counts = torch.zeros(W, dtype=torch.int32, device="cuda")
    L = abi.lib()
    ms_counts = timed(lambda: abi.check(L.msq_insertion_counts(ctx.h, dr.view, b0, W, counts.data_ptr())))
    total = int(device.insertion_counts(ctx, dr, b0, W).sum())
    t0 = time.perf_counter()
    recs = device.insertion_records(ctx, dr, b0, W, capacity=total)
    t_rec = time.perf_counter() - t0
    nuc, codon = device.alloc_counts(ctx, W)
    ms_pileup_plain = timed(lambda: device.pileup(ctx, dr, thr, b0, W, nuc=nuc, codon=codon, check=False))
    dr_n = device.upload(device.pack_device(ctx, reads))
    ms_pileup_norm = timed(lambda: device.pileup(ctx, dr_n, thr, b0, W, nuc=nuc, codon=codon, check=False))
    bytes_planes = dr.n_groups * 16
    out = {"what": "insertion scan on plain-layout reads", "workload": wl.name, "reads": a.reads, "columns": W,
           "groups": dr.n_groups, "insertions": total, "records": len(recs),
           "insertion_counts_ms": ms_counts, "insertion_counts_GBps": bytes_planes / ms_counts / 1e6,
           "insertion_records_all_columns_s_incl_python_decode": t_rec,
           "pileup_plain_layout_ms": ms_pileup_plain, "pileup_normalized_ms": ms_pileup_norm,
           "device_pack_s_incl_python_marshalling": t_pack}
    print(json.dumps(out), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
