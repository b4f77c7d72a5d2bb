#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: calls, average ns, kernel name."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%6s %12.0f  %s" % (r["Calls"], float(r["AverageNs"]), r["Name"][:70]))
