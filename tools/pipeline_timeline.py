#!/usr/bin/env python3
"""Start / end of every kernel of a pipelined run from a rocprofv3 --kernel-trace CSV: do the eigen kernel of window w + 1 and the
boundary-condition kernel of window w overlap?   python3 tools/pipeline_timeline.py <kernel_trace.csv> [first_rows]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        m = re.search(r"rtd_\w+(<[^>]*>)?", r["Kernel_Name"])
        if m:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0 = rows[0][0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = len(rows) // 2
for s, e, name, q, st in rows[skip:skip + n]:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  queue {q} stream {st}  {name}")
