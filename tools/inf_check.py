#!/usr/bin/env python3
"""Device inflate of a synthetic BAM (tools/bench_bam.py's writer) against zlib, member by member; prints the first difference."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench_bam
from minorseq_amd import abi, device
from test_gpu_bgzf import inflate_device, scan

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
path = "/tmp/inf_check.bam"
bench_bam.write(path, n, 2843, False)
blob = open(path, "rb").read()
members = scan(blob)
ctx = device.Ctx(0)
try:
    got = inflate_device(ctx, blob, members)
except abi.MsqError as e:
    print("ERROR", e)
    got = None
bad = 0
for i, m in enumerate(members):
    want = zlib.decompress(blob[m[0]:m[0] + m[2]], -15)
    if got is None:
        try:
            g = inflate_device(ctx, blob, [(m[0], 0, m[2], m[3])])
        except abi.MsqError as e:
            print("member", i, "refused:", e)
            bad += 1
            if bad > 3:
                break
            continue
    else:
        g = got[m[1]:m[1] + m[3]]
    if g != want:
        k = next(j for j in range(len(want)) if g[j] != want[j])
        print("member", i, "differs at", k, "of", len(want), "want", want[k - 4:k + 8].hex(), "got", g[k - 4:k + 8].hex())
        bad += 1
        if bad > 3:
            break
print("members", len(members), "bad", bad)
