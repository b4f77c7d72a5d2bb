#!/usr/bin/env python3
"""What a GPU inflate for the BAM front would pay at the boundary: the compressed file image (pageable, mapped) to HBM
and the inflated stream back into pageable host memory, at the sizes of tools/bench_bam.py's 10^5-read file
(287 MB in, ~600 MB out), against pinned memory.  Prints one JSON line."""
import json
import time

import numpy as np
import torch


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    n_in, n_out = 287 << 20, 600 << 20
    dev = torch.device("cuda", 0)
    d_in = torch.empty(n_in, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n_out, dtype=torch.uint8, device=dev)
    h_in = torch.from_numpy(np.random.randint(0, 255, n_in, dtype=np.uint8))
    out = {}
    out["h2d_pageable_ms"] = 1e3 * timed(lambda: d_in.copy_(h_in))
    h_pin = torch.empty(n_in, dtype=torch.uint8).pin_memory()
    out["h2d_pinned_ms"] = 1e3 * timed(lambda: d_in.copy_(h_pin, non_blocking=True))
    t0 = time.perf_counter()
    h_pin.copy_(h_in)
    out["host_copy_into_pinned_ms"] = 1e3 * (time.perf_counter() - t0)

    def d2h_fresh():  # into memory never touched before, as the front's inflated stream is
        h = torch.empty(n_out, dtype=torch.uint8)
        h.copy_(d_out)
    out["d2h_pageable_first_touch_ms"] = 1e3 * timed(d2h_fresh, reps=2)
    h_old = torch.zeros(n_out, dtype=torch.uint8)
    out["d2h_pageable_touched_ms"] = 1e3 * timed(lambda: h_old.copy_(d_out))
    h_pin_out = torch.empty(n_out, dtype=torch.uint8).pin_memory()
    out["d2h_pinned_ms"] = 1e3 * timed(lambda: h_pin_out.copy_(d_out, non_blocking=True))
    out["bytes_in"], out["bytes_out"] = n_in, n_out
    print(json.dumps(out))


if __name__ == "__main__":
    main()
