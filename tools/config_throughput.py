#!/usr/bin/env python3
"""Throughput of the HIP path on the other BASELINE configs (cfg3 big/small, cfg5) -- diagnostic, not bench.py."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd
from pydisort_amd import synthetic

for name, maker, kw, C in [("cfg3_small(L6,Q8)", synthetic.cfg3_columns, dict(big=False), 4096),
                           ("cfg3_big(L8,Q16)", synthetic.cfg3_columns, dict(big=True), 4096),
                           ("cfg4(L20,Q32)", synthetic.cfg4_columns, {}, 1024),
                           ("cfg5(L50,Q64)", synthetic.cfg5_columns, {}, 128)]:
    cfg = maker(C, **kw)
    _, sol = pydisort_amd.pydisort_batch(**cfg)
    plan = sol.plan
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0, np.pi / 2, np.pi]))
    plan.run(); plan.synchronize()
    plan.enable_timing(True); plan.timing(reset=True)
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        plan.run()
    plan.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = plan.timing()
    print(f"{name:20s} C={C:5d}  {C / dt:10.1f} col/s   stage ms:", {k: round(v[0] / max(v[1], 1), 2) for k, v in st.items()},
          "sweeps", plan.max_sweeps(), "pivoted chains", plan.pivoted_chains(), "of", C * plan.M, flush=True)
    plan.close()
