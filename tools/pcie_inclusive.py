#!/usr/bin/env python3
"""PCIe-inclusive rate of the batch API: host arrays in (prepared on the host), host arrays out, one call.
Reported in DESIGN.md next to (never instead of) bench.py's HBM-resident `value`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd
from pydisort_amd import synthetic

C = 2048
cfg = synthetic.cfg4_columns(C)
tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
phi = np.array([0.0, np.pi / 2, np.pi])
pydisort_amd.pydisort_batch(**synthetic.cfg4_columns(8))  # warm the runtime
for rep in range(3):
    t0 = time.perf_counter()
    _, sol = pydisort_amd.pydisort_batch(**cfg)   # host prep + plan creation (hipMalloc) + upload + solve
    t1 = time.perf_counter()
    u = sol.u(tau, phi)
    fu = sol.flux_up(tau)
    fd = sol.flux_down(tau)
    t2 = time.perf_counter()
    print(f"rep {rep}: prep+alloc+upload+solve {t1 - t0:.3f} s, evaluate+download {t2 - t1:.3f} s, "
          f"end-to-end {C / (t2 - t0):.0f} column-solves/s (C = {C})", flush=True)
    sol.plan.close()
