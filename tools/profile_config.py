#!/usr/bin/env python3
"""The workload of one BASELINE config as a bare program for rocprofv3 (every kernel launch in the trace is the workload):
    python3 tools/profile_config.py cfg5 128 0 3      # config, columns, columns per window (0: one window), passes
Prints the resident rate and the HIP-event stage times of the last pass."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd  # noqa: E402
from pydisort_amd import synthetic  # noqa: E402

name, C, window, passes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
maker, kw = {"cfg3_small": (synthetic.cfg3_columns, dict(big=False)), "cfg3_big": (synthetic.cfg3_columns, dict(big=True)),
             "cfg4": (synthetic.cfg4_columns_block, {}), "cfg5": (synthetic.cfg5_columns, {}),
             "cfg4_cloud": (synthetic.cfg4_cloud_columns, {})}[name]  # (cfg4_cloud: a layer with omega = 1 - 1e-6 in every column)
cfg = maker(C, **kw)
_, sol = pydisort_amd.pydisort_batch(work_columns=window, _defer_solve=True, **cfg)
plan = sol.plan
tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
plan.set_eval_points(tau, np.array([0.0, np.pi / 2, np.pi]))
plan.run()
plan.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    plan.run()
plan.synchronize()
dt = (time.perf_counter() - t0) / passes
if "--timing" in sys.argv:
    plan.enable_timing(True)
    plan.timing(reset=True)
    plan.run()
    st = plan.timing()
    print({k: round(v[0] / max(v[1], 1), 3) for k, v in st.items()})
print(f"{name} C={C} windows={plan.windows()} {C / dt:.1f} col/s  max sweeps {plan.max_sweeps()} pivoted chains {plan.pivoted_chains()}")
plan.close()
