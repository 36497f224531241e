#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per (kernel, grid size).
Launches of the same kernel on different problem sizes (bench.py's parity and only_flux legs) stay separate."""
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            name = (re.search(r"rtd_\w+", row["Kernel_Name"]) or re.search(r"\w+", row["Kernel_Name"])).group(0)
            if not name.startswith("rtd_"):
                continue
            acc[(name, int(row["Grid_Size"]), int(row.get("VGPR_Count", 0) or 0))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, g, v), d in sorted(acc.items()):
    print(k, "grid=%d vgpr=%d" % (g, v), {c: round(sum(x) / len(x), 1) for c, x in sorted(d.items())},
          "n=%d" % len(next(iter(d.values()))))
