#!/usr/bin/env python3
"""Latency of the drop-in pydisort() for ONE column (reference: TP5 30 ms, cfg4 154 ms on one CPU core)."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import goldens, pydisort_amd
from pydisort_amd import synthetic
warnings.simplefilter("ignore")
cases = {"TP5a (L1,Q48,NT)": goldens.load("5a")[0]["kwargs"], "TP9c (L6,Q8)": goldens.load("9c")[0]["kwargs"],
         "cfg4 column (L20,Q32)": synthetic.column_kwargs(synthetic.cfg4_columns(1), 0)}
pydisort_amd.pydisort(**cases["TP9c (L6,Q8)"])
for name, kw in cases.items():
    ts, te = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        res = pydisort_amd.pydisort(**kw)
        res[1](0.5 * np.atleast_1d(kw["tau_arr"])[-1])   # forces completion of the solve
        t1 = time.perf_counter()
        tau = np.linspace(0, np.atleast_1d(kw["tau_arr"])[-1], 21)
        res[4](tau, np.array([0.0, 1.0, 2.0]))
        t2 = time.perf_counter()
        ts.append(t1 - t0); te.append(t2 - t1)
    print(f"{name:24s} solve {1e3 * min(ts):7.2f} ms   evaluate u(21 tau, 3 phi) {1e3 * min(te):6.2f} ms", flush=True)
