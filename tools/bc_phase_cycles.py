#!/usr/bin/env python3
"""Formats the phase stamps printed by a -DRTD_BCF_STAMPS build of rtd_bc_mfma_kernel (cycles per phase and layer).

Usage (GPU box):  RTD_LIB=<stamped librtd.so> python tools/bc_phase_cycles.py [columns]
Build the stamped library with  RTD_EXTRA_FLAGS=-DRTD_BCF_STAMPS python pythonic-disort_amd/build.py --force  (into a copy).
With 16 columns every SIMD holds at most one wavefront: the unloaded latencies; with 2048 the benchmark's contention."""
import collections
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
import numpy as np
sys.path[:0] = [%r, %r]
import pydisort_amd
from pydisort_amd import synthetic
C = int(sys.argv[1])
cfg = synthetic.cfg4_columns(C)
_, sol = pydisort_amd.pydisort_batch(**cfg, _defer_solve=True)
plan = sol.plan
tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
plan.set_eval_points(tau, np.array([0.0, np.pi / 2, np.pi]))
plan.run(); plan.synchronize()
''' % (ROOT, os.path.join(ROOT, "pythonic-disort_amd"))

if __name__ == "__main__":
    C = sys.argv[1] if len(sys.argv) > 1 else "2048"
    # (three stamps per forward iteration: the kernel forms its products in one block with the carry, no boundary between them)
    out = subprocess.run([sys.executable, "-c", CHILD, C], capture_output=True, text=True).stdout
    d = collections.defaultdict(dict)
    for ln in out.splitlines():
        if ln.startswith("ST "):
            _, cm, i, v = ln.split()
            d[int(cm)][int(i)] = int(v)
    L = 20
    for cm, v in sorted(d.items()):
        seq = [v[i] for i in range(1, max(v) + 1)]
        fw = np.array(seq[1:1 + 3 * (L - 1)]).reshape(L - 1, 3)
        rest = seq[1 + 3 * (L - 1):]
        bw = np.array(rest[3:3 + 2 * (L - 1)]).reshape(L - 1, 2)
        print(f"chain {cm}: total {sum(seq)} cycles; prologue {seq[0]}")
        print("  forward, mean per layer [top+loads, elimination, wait+stores+products+rho+carry]:", fw.mean(0).round(0))
        print("  last elimination, bottom boundary, first backward part:", rest[:3])
        print("  backward, mean per layer [operands+C+, rest]:", bw.mean(0).round(0), " tail:", rest[3 + 2 * (L - 1):])
