#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel (substring filter)."""
import csv
import collections
import sys


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if filt and filt not in name:
            continue
        acc[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        print(name)
        for c, v in sorted(cs.items()):
            print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
