#!/usr/bin/env python3
"""Rate of the 32-stream path on a batch with a conservative cloud layer (omega = 1 - 1e-6) in every column, beside the plain
cfg4 batch: what the pivoted elimination of the near-conservative mode-0 chains costs.
    python3 tools/cloud_batch.py [columns] [passes]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd  # noqa: E402
from pydisort_amd import synthetic  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for name, cfg in (("cfg4", synthetic.cfg4_columns_block(C)), ("cfg4 + cloud layer in every column", synthetic.cfg4_cloud_columns(C))):
    _, sol = pydisort_amd.pydisort_batch(work_columns=256, _defer_solve=True, **cfg)
    plan = sol.plan
    plan.set_eval_points(np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1), np.array([0.0, np.pi / 2, np.pi]))
    plan.run()
    plan.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        plan.run()
    plan.synchronize()
    dt = (time.perf_counter() - t0) / passes
    out = plan.fetch()
    print(f"{name}: {C / dt:.0f} col/s ({C} columns, windows of 256), finite: {bool(np.all(np.isfinite(out['u'])))}, max sweeps {plan.max_sweeps()}")
    plan.close()
