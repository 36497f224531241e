#!/usr/bin/env python3
"""Experiment: do two plans (own HIP streams) overlap the VALU-bound eigen kernel of one half-batch with the
HBM-bound BC/eval kernels of the other?  Compares 1 x 2048 columns against 2 x 1024 and 4 x 512 run concurrently."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
from pydisort_amd import synthetic
from pydisort_amd._engine import Plan
from pydisort_amd._prepare import prepare_columns

def make_plan(C, first):
    cfg = synthetic.cfg4_columns(C, first=first)
    N = 16
    prep = prepare_columns(cfg["tau_arr"], cfg["omega_arr"], 32, cfg["Leg_coeffs_all"], cfg["mu0"], cfg["I0"], cfg["phi0"], 32, 32,
                           np.zeros((C, N, 32)), np.zeros((C, N, 32)), cfg["f_arr"], np.zeros((C, 20, 0)),
                           np.zeros((C, 0, N, N)), np.zeros((C, 0, N)))
    plan = Plan(prep)
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0, np.pi / 2, np.pi]))
    return plan

TOTAL = 2048
for nplans in (1, 2, 4):
    C = TOTAL // nplans
    plans = [make_plan(C, i * C) for i in range(nplans)]
    for p in plans:
        p.run()
    for p in plans:
        p.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for step in range(5):
            for p in plans:      # launches are asynchronous: the streams run concurrently
                p.run()
        for p in plans:
            p.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    print(f"{nplans} plan(s) x {C} columns: {1e3 * best:.2f} ms per {TOTAL} columns -> {TOTAL / best:.0f} column-solves/s", flush=True)
    for p in plans:
        p.close()
