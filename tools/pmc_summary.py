#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel per dispatch."""
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[(re.search(r"rtd_\w+", row["Kernel_Name"]) or re.search(r"\w+", row["Kernel_Name"])).group(0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())}, "n=%d" % len(next(iter(d.values()))))
