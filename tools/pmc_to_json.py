#!/usr/bin/env python3
"""Turns the per-pass summaries of tools/pmc_summary.py (one text file per rocprofv3 --pmc pass) into the JSON that
bench.py reads for roofline.traffic: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (both counters are in
KiB; FETCH_SIZE counts half of the bytes of coalesced streaming reads on gfx950: MI355X_MICROARCH.md, HBM section).

Usage: python tools/pmc_to_json.py out.json "description" [--columns-per-launch N] pass1.txt pass2.txt ..."""
import ast
import json
import re
import sys

out, desc, files = sys.argv[1], sys.argv[2], sys.argv[3:]
cols = None
if files and files[0] == "--columns-per-launch":
    cols, files = int(files[1]), files[2:]
kern = {}
for path in files:
    for line in open(path):
        m = re.match(r"(rtd_\w+) grid=(\d+) vgpr=(\d+) (\{.*\}) n=(\d+)", line)
        if not m:
            continue
        # a kernel launched on several grid sizes in one run (the short last window of a batch): the LARGEST grid -- the full
        # window -- is the one reported, with its own launch count
        grid, n = int(m.group(2)), int(m.group(5))
        k = kern.get(m.group(1))
        if k is None or grid > k["grid"]:
            k = kern[m.group(1)] = {**{c: v for c, v in (k or {}).items() if False}, "grid": grid, "launches_averaged": n}
        elif grid < k["grid"]:
            continue
        k.update(ast.literal_eval(m.group(4)))
total = 0.0
for name, k in kern.items():
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        k["hbm_bytes_per_launch"] = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        if not name.startswith("rtd_tables"):  # the table kernels run once per change of the inputs, not per window
            total += k["hbm_bytes_per_launch"]
json.dump({"source": desc,
           "correction": "FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE doubled (gfx950 reports half of the bytes of "
                         "coalesced streaming reads); separate --pmc passes for FETCH_SIZE and WRITE_SIZE",
           "kernels": kern, "total_hbm_bytes_per_step": total, "columns_per_launch": cols}, open(out, "w"), indent=1, sort_keys=True)
print(out, "total GB per step: %.2f" % (total / 1e9))
