#!/usr/bin/env python3
"""Host-to-host cost of a batch call next to the device time of the same pass (run on the GPU box).

For each workload: plan.run() + synchronize (results stay in HBM), run_fetch into preallocated / fresh arrays, and one whole
pydisort_batch() call from NumPy inputs (input checks, plan creation, uploads, solve, evaluation at the interfaces, download) with
the cProfile entries that cost most.  Usage:  python tools/host_overhead.py [profile]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd  # noqa: E402
from pydisort_amd import synthetic  # noqa: E402

WORK = [("cfg3 L6/Q8 x 1024", "cfg3_columns", dict(big=False), 1024, 1024),
        ("cfg3 L8/Q16 x 1024", "cfg3_columns", dict(big=True), 1024, 1024),
        ("cfg4 x 4096", "cfg4_columns", {}, 4096, 256),
        ("cfg5 x 256", "cfg5_columns", {}, 256, 128)]


def med(fn, n=15):
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return 1e3 * sorted(t)[len(t) // 2]


def main():
    prof = len(sys.argv) > 1 and sys.argv[1] == "profile"
    for name, maker, kw, C, win in WORK:
        cfg = getattr(synthetic, maker)(C, **kw)
        tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
        phi = np.array([0.0, np.pi / 2, np.pi])
        _, sol = pydisort_amd.pydisort_batch(work_columns=win, _defer_solve=True, **cfg)
        plan = sol.plan
        plan.set_eval_points(tau, phi)
        plan.run(); plan.synchronize()

        def resident():
            plan.run(); plan.synchronize()
        out = plan.run_fetch()
        t_res, t_pre, t_fresh = med(resident), med(lambda: plan.run_fetch(out)), med(plan.run_fetch)
        plan.close()

        def whole():
            _, s = pydisort_amd.pydisort_batch(work_columns=win, **cfg)
            r = s.u(tau, phi), s.u0(tau), s.flux_up(tau), s.flux_down(tau)
            s.plan.close()
            return r
        whole()
        t_whole = med(whole, 7)
        spans = []
        for _ in range(7):
            t = [time.perf_counter()]
            _, s = pydisort_amd.pydisort_batch(work_columns=win, **cfg); t.append(time.perf_counter())
            s.u(tau, phi); t.append(time.perf_counter())
            s.u0(tau); t.append(time.perf_counter())
            s.flux_up(tau); t.append(time.perf_counter())
            s.flux_down(tau); t.append(time.perf_counter())
            s.plan.close(); t.append(time.perf_counter())
            spans.append(np.diff(t) * 1e3)
        spans = np.array(spans)
        print("    pydisort_batch / u / u0 / flux_up / flux_down / close, ms, 7 calls:\n" + "\n".join("      " + " ".join(f"{v:9.3f}" for v in r) for r in spans), flush=True)
        print(f"{name:22s} resident {t_res:8.3f} ms   run_fetch(prealloc) {t_pre:8.3f}   run_fetch(fresh) {t_fresh:8.3f}   "
              f"pydisort_batch + 4 closures {t_whole:8.3f} ms   ({C / t_whole * 1e3:,.0f} col/s host to host)", flush=True)
        if prof:
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(5):
                whole()
            pr.disable()
            st = pstats.Stats(pr, stream=sys.stdout)
            st.sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
