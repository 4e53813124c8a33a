#!/usr/bin/env python3
"""Compact view of a rocprofv3 *kernel_stats.csv: short kernel name, calls, average / min microseconds, share."""
import csv
import re
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"^void ", "", r["Name"])
    n = re.sub(r"\(.*", "", n)[:60]
    print("%-60s %6s calls  avg %8.2f us  min %8.2f  %5.1f %%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["Percentage"])))
