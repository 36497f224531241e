#!/usr/bin/env python3
"""Stage times of the 66 ... 128-stream path (NP = 64 instances: rtd_eigen_kernel<64, 2>, rtd_iface_mfma_kernel,
rtd_sweep_wide_kernel, rtd_eval_kernel<64>) on synthetic Henyey-Greenstein columns -- diagnostic, not bench.py.
Usage: python tools/many_stream_timing.py [columns]      (RTD_BC_WIDE_V1=1: the row-per-lane BC kernels of rounds 1-3)
Prints, per configuration (NQuad, layers, Fourier modes, columns), the plan's per-stage HIP-event times in ms and columns/s."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd
from pydisort_amd import synthetic
warnings.simplefilter("ignore")
C0 = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for NQuad, L, M, C in [(128, 50, 64, C0), (96, 20, 48, 4 * C0), (72, 50, 36, 2 * C0)]:
    cfg = synthetic.cfg4_columns(C, L=L, NQuad=NQuad, g_hi=0.9)
    cfg["NFourier"] = M
    _, sol = pydisort_amd.pydisort_batch(_defer_solve=True, **cfg)
    plan = sol.plan
    tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    plan.set_eval_points(tau, np.array([0.0, 1.0]))
    plan.run(); plan.synchronize()
    t0 = time.perf_counter()
    plan.run(); plan.synchronize()
    dt = time.perf_counter() - t0
    plan.enable_timing(True); plan.timing(reset=True); plan.run(); st = plan.timing(reset=True)
    print(f"NQuad {NQuad} L {L} M {M} C {C}: {C / dt:8.1f} col/s  stage ms", {k: round(v[0], 3) for k, v in st.items() if v[0] > 0.02}, flush=True)
    plan.close()
