#!/usr/bin/env python3
"""Median s_memtime cycles per phase of the fused eigen kernel from the EIGSTAMP lines of a -DRTD_EIG_STAMPS build
(tools/build_variant.py stamps -DRTD_EIG_STAMPS; RTD_LIB=.../librtd_stamps.so python tools/profile_config.py cfg5 128 0 1 | python tools/eig_phase_cycles.py)."""
import collections
import re
import statistics
import sys

rows = collections.defaultdict(list)
for line in sys.stdin:
    if not line.startswith(("EIGSTAMP", "T2STAMP")):
        continue
    m_np, m_sw = re.search(r"np (\d+)", line), re.search(r"sweeps (\d+)", line)
    if not (m_np and m_sw and ":" in line):  # (interleaved device printf output)
        continue
    np_, sweeps = int(m_np.group(1)), int(m_sw.group(1))
    vals = {k: int(v) for k, v in re.findall(r"(\S+) (-?\d+)", line.split(":", 1)[1])}
    vals["sweeps"] = sweeps
    if "total" in vals:  # (device printf output of concurrent wavefronts can interleave: incomplete lines are dropped)
        rows[np_].append(vals)
for np_, rs in sorted(rows.items()):
    keys = [k for k in rs[0] if k != "sweeps"]
    print(f"NP = {np_}: {len(rs)} wavefronts sampled, mean sweeps {statistics.mean(r['sweeps'] for r in rs):.2f}")
    tot = statistics.median(r["total"] for r in rs)
    for k in keys:
        med = statistics.median(r[k] for r in rs)
        print(f"  {k:12s} median {med:9.0f} cycles  ({100.0 * med / tot:5.1f} % of the median total)   mean {statistics.mean(r[k] for r in rs):9.0f}")
