"""Per-workgroup timing of the stream kernel (library built with EXTRA=-DMSQ_WGT, MSQ_LIB_DIR=<its dir>): when each
workgroup of the last launch started, left its batch loop and ended -- the kernel ends with its slowest workgroup.
usage: MSQ_LIB_DIR=lib_wgt python tools/wgt.py [workload] [reads] [minor args as for synth.workload: k=v ...]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from minorseq_amd import abi, device, synth

name = sys.argv[1] if len(sys.argv) > 1 else "pol_1m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v else int(v)
ctx = device.Ctx(0)
wl = synth.workload(name, n_reads=n, **kw)
dr = device.synth_device(ctx, wl, 0, n)
L = abi.lib()
thr = abi.QvThresholds.make()
got = torch.zeros(wl.W * 70, dtype=torch.int32, device="cuda")
gb = torch.zeros(2, dtype=torch.int32, device="cuda")
buf = (ctypes.c_ulonglong * 6144)()
for rep in range(3):
    for _ in range(150):
        abi.check(L.msq_pileup_step(ctx.h, dr.view, thr, wl.win_begin0, wl.W, gb.data_ptr(), got.data_ptr()))
    abi.check(L.msq_pileup_check(ctx.h))
    ctx.sync()
    assert L.msq_debug_wgt(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(6, 1024)
    nwg = int((a[1] > 0).sum())
    t0 = a[0, :nwg].min()
    start, end, loop = a[0, :nwg] - t0, a[1, :nwg] - t0, a[2, :nwg] - t0
    q = lambda x: "min %6.1f med %6.1f p90 %6.1f max %6.1f" % (x.min() / 100, np.median(x) / 100, np.percentile(x, 90) / 100, x.max() / 100)
    print("launch of %d workgroups (us): start %s | loop end %s | end %s" % (nwg, q(start), q(loop), q(end)))
    setup, first, guide = a[3, :nwg] - a[0, :nwg], a[4, :nwg] - a[3, :nwg], a[5, :nwg] - a[4, :nwg]
    print("   wave 0 of each workgroup: set-up %s | first batch %s | guide %s | flush %s" % (q(setup), q(first), q(guide), q(end - loop)))
    print("   loop end by wg %% 8 (XCD): " + " ".join("%.1f" % (loop[i::8].mean() / 100) for i in range(8)))
    o = np.argsort(loop)
    print("   slowest loops: " + " ".join("%d:%.1f" % (i, loop[i] / 100) for i in o[-6:]) + " | fastest: " + " ".join("%d:%.1f" % (i, loop[i] / 100) for i in o[:4]))
ctx.close()
