"""Time msq_reads_partition (pack-time layout for wide windows) on a synthetic workload: python tools/bench_partition.py
[--workload hiv_mix_200k] [--reads N] -> one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hiv_mix_200k")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--slab-columns", type=int, default=0)
    a = ap.parse_args()
    import torch
    from minorseq_amd import device, synth
    wl = synth.workload(a.workload)
    n = a.reads or wl.n_reads
    ctx = device.Ctx(0)
    dr = device.synth_device(ctx, wl, 0, n)
    b0, e0 = device.window_bounds(ctx, dr)
    times = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        part = device.Partition(ctx, dr, b0, e0 - b0, a.slab_columns)
        ctx.sync()
        times.append(time.perf_counter() - t0)
        groups = part.n_groups() if callable(getattr(part, "n_groups", None)) else None
        slabs = part.n_slabs
        part.close()
    print(json.dumps({"what": "msq_reads_partition", "workload": wl.name, "reads": n, "columns": e0 - b0, "slabs": slabs,
                      "source_groups": dr.n_groups, "partition_groups": groups, "first_s": times[0], "best_s": min(times[1:])}))


if __name__ == "__main__":
    main()
