"""Debug aid: run one edge set (tests/cases.py) or an ad-hoc variant through the C ABI and print GPU vs oracle counts."""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import cases
from minorseq_amd import abi, device
from oracle import orc
ctx = device.Ctx(0)
none = (-1, -1, -1, -1)
extra = {
    "x_first": (cases.make_reads([(5, "=X====", "ATGTGG"), (5, "======", "ACGTGG"), (5, "======", "ACGTGG"), (5, "======", "ACGTGG"), (5, "======", "ACGTGG")]), none),
    "x_last": (cases.make_reads([(5, "======", "ACGTGG"), (5, "======", "ACGTGG"), (5, "======", "ACGTGG"), (5, "======", "ACGTGG"), (5, "=X====", "ATGTGG")]), none),
    "x_last8": (cases.make_reads([(5, "======", "ACGTGG")] * 7 + [(5, "=X====", "ATGTGG")]), none),
}
for name in sys.argv[1:]:
    reads, thr = extra[name] if name in extra else cases.edge_sets()[name]
    m = orc.Msa(reads, thr)
    dr = device.upload(device.pack_host(reads), ctx.device)
    b0, e0 = device.window_bounds(ctx, dr)
    W = e0 - b0
    nuc, codon = device.pileup(ctx, dr, abi.QvThresholds.make(*thr), b0, W)
    g = nuc.cpu().numpy()[:W*6].reshape(W, 6); o = m.nuc_counts()
    gc = codon.cpu().numpy()[:W*64].reshape(W, 64); oc = m.codon_counts()
    print(name, "W", W, "nuc equal", (g == o).all(), "codon equal", (gc == oc).all())
    if not (g == o).all():
        print("gpu\n", g.T, "\norc\n", o.T)
    for c in range(W):
        if (gc[c] != oc[c]).any():
            print("col", c, "gpu", {k: int(v) for k, v in enumerate(gc[c]) if v}, "orc", {k: int(v) for k, v in enumerate(oc[c]) if v})
