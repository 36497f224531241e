#!/usr/bin/env python3
"""Where the host-to-host time of BASELINE's 10^5-column batch goes (bench.py `e2e`): the stages of solve_columns_streamed timed one by
one (run on the GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd  # noqa: E402
from pydisort_amd import synthetic  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
cfg = synthetic.cfg4_columns_block(C, first=0)
tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
phi = np.array([0.0, np.pi / 2, np.pi])
out = dict(u=np.zeros((C, 32, 21, 3)), u0=np.zeros((C, 32, 21)), flux_up=np.zeros((C, 21)), flux_down_diffuse=np.zeros((C, 21)),
           flux_down_direct=np.zeros((C, 21)))
for a in out.values():
    a.fill(0.0)
small = {k: (v[:512] if isinstance(v, np.ndarray) else v) for k, v in cfg.items()}
pydisort_amd.solve_columns_streamed(small, tau[:512], phi, chunk_columns=256)
for rep in range(3):
    t = [time.perf_counter()]
    _, sol = pydisort_amd.pydisort_batch(work_columns=256, device_prepare=True, _defer_solve=True, **cfg); t.append(time.perf_counter())
    sol.plan.synchronize(); t.append(time.perf_counter())
    sol._tau(tau); t.append(time.perf_counter())
    sol.plan.set_eval_points(tau, phi); t.append(time.perf_counter())
    sol.plan.run_fetch(out); t.append(time.perf_counter())
    sol.plan.close(); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print(f"checks + plan + upload {d[0]:7.1f} | drain {d[1]:6.1f} | tau range check {d[2]:6.1f} | eval points {d[3]:6.1f} | run_fetch {d[4]:7.1f} | "
          f"close {d[5]:5.1f} | total {sum(d):7.1f} ms", flush=True)
    t0 = time.perf_counter()
    pydisort_amd.solve_columns_streamed(cfg, tau, phi, chunk_columns=256, out=out)
    print(f"   solve_columns_streamed: {(time.perf_counter() - t0) * 1e3:7.1f} ms", flush=True)
_, sol = pydisort_amd.pydisort_batch(work_columns=256, device_prepare=True, _defer_solve=True, **cfg)
sol.plan.set_eval_points(tau, phi)
sol.plan.run(); sol.plan.synchronize()
t0 = time.perf_counter()
sol.plan.run(); sol.plan.synchronize()
print(f"   resident pass (run + synchronize): {(time.perf_counter() - t0) * 1e3:7.1f} ms")
