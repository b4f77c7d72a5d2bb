#!/usr/bin/env python3
"""Extract DATA (not code) from the reference's predefined target configs into minorseq_amd/data/.

The reference embeds three JSON target configs (the same JSON a user passes with `juliet -c`) as string
literals at src/TargetConfig.cpp:257-264.  From the "HIV" one this script keeps what the hot path consumes:
the gene list (name, begin, end: HXB2 coordinates, public knowledge) and the HXB2 reference sequence
(GenBank K03455, public).  DRM tables and everything else are left behind.  Run in the build container
only (the reference is absent on the GPU box); outputs are committed.
"""
import json
import os
import re
import sys

REF = os.environ.get("MINORSEQ_REF", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "minorseq_amd", "data")


def main():
    src = open(os.path.join(REF, "src", "TargetConfig.cpp"), encoding="utf-8").read()
    configs = {}
    for m in re.finditer(r'\{"([\w-]+)",\s*R"\((\{.*?\})\)"\}', src, re.S):
        configs[m.group(1)] = json.loads(m.group(2))
    hiv = configs["HIV"]
    seq = hiv["referenceSequence"]
    assert set(seq) <= set("ACGT") and len(seq) == 9719, len(seq)
    with open(os.path.join(OUT, "hxb2.fasta"), "w") as f:
        f.write(">HIV_HXB2 K03455 %d nt\n" % len(seq))
        for i in range(0, len(seq), 70):
            f.write(seq[i:i + 70] + "\n")
    genes = [{"name": g["name"], "begin": g["begin"], "end": g["end"]} for g in hiv["genes"]]
    with open(os.path.join(OUT, "hiv_genes.json"), "w") as f:
        json.dump({"referenceName": hiv.get("referenceName", ""), "genes": genes}, f, indent=1)
    abl = configs["ABL1"]
    with open(os.path.join(OUT, "abl1_genes.json"), "w") as f:
        json.dump({"genes": [{"name": g["name"], "begin": g["begin"], "end": g["end"]} for g in abl["genes"]]}, f, indent=1)
    print("genes:", len(genes), "ref:", len(seq))


if __name__ == "__main__":
    sys.exit(main())
