#!/usr/bin/env python3
"""Event counts of the aligned-set pileup kernel for one pass over a synthetic workload (library built with
-DMSQ_ROWS_EXP=4: make LIBDIR=../lib_dbg EXTRA=-DMSQ_ROWS_EXP=4; run with MSQ_LIB_DIR=lib_dbg).
usage: tools/rows_counters.py [workload] [reads] [synth key=value ...]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from minorseq_amd import abi, device, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "pol_1m"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    kw = {}
    for a in sys.argv[3:]:
        k, v = a.split("=")
        kw[k] = float(v) if "." in v else int(v)
    wl = synth.workload(name, n_reads=n, **kw)
    ctx = device.Ctx(0)
    dr = device.synth_device(ctx, wl, 0, n)
    al = device.align(ctx, dr, abi.QvThresholds.make(), wl.win_begin0)
    out = (C.c_ulonglong * 8)()
    abi.lib().msq_debug_rows_counters(out)
    nuc, cod = device.pileup(ctx, al, abi.QvThresholds.make(), wl.win_begin0, wl.W)
    torch.cuda.synchronize()
    abi.check(abi.lib().msq_debug_rows_counters(out))
    names = ["entries", "drains", "nuc_atomics", "codon_atomics", "lds_alt", "lds_sub"]
    print(name, n, kw, {k: int(out[i]) for i, k in enumerate(names)}, "per read:", {k: round(out[i] / n, 4) for i, k in enumerate(names)})


if __name__ == "__main__":
    main()
