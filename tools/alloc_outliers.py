#!/usr/bin/env python3
"""How long creating + destroying a plan takes, call after call (run on the GPU box): a retained 4 096-column cfg4 plan is 14 GB of
device memory, which every pydisort_batch() call used to hipMalloc and hipFree.  Three legs, each in its own process:
the library with its pool of large blocks off (RTD_POOL_BYTES=0), the library as it is, and bare hipMalloc / hipFree of the same
sizes -- the stalls are the runtime's (lazy reclaim of freed memory), the pool is what keeps them out of a serving loop."""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
N = 40


def summary(what, t_new, t_close=None):
    late = [(i, round(v)) for i, v in enumerate(t_new) if v > 50]
    s = f"{what}: create ms median {np.median(t_new):.2f} max {max(t_new):.1f}; calls over 50 ms (index, ms): {late}"
    if t_close:
        s += f"; destroy ms median {np.median(t_close):.2f} max {max(t_close):.1f}"
    print(s, flush=True)


def plans():
    import pydisort_amd
    from pydisort_amd import synthetic
    from pydisort_amd._engine import Plan
    cfg = synthetic.cfg4_columns(4096)
    _, sol = pydisort_amd.pydisort_batch(work_columns=256, _defer_solve=True, **cfg)
    prep = sol.plan.prep
    sol.plan.close()
    pydisort_amd.pool_trim()
    for retain in (-1, 0):
        t_new, t_close, nbytes = [], [], 0
        for _ in range(N):
            t0 = time.perf_counter()
            p = Plan(prep, 0, 256, retain)
            t1 = time.perf_counter()
            nbytes = p.device_bytes()
            p.close()
            t_close.append((time.perf_counter() - t1) * 1e3)
            t_new.append((t1 - t0) * 1e3)
        summary(f"  plan of {nbytes / 1e9:.2f} GB ({'retained' if retain else 'windowed'}), pool holds {pydisort_amd.pool_bytes() / 1e9:.2f} GB", t_new, t_close)


def bare():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    for gb in (14.0, 3.0):
        t_new = []
        for _ in range(N):
            ptr = ctypes.c_void_p()
            t0 = time.perf_counter()
            rc = hip.hipMalloc(ctypes.byref(ptr), int(gb * 1e9))
            t_new.append((time.perf_counter() - t0) * 1e3)
            assert rc == 0
            hip.hipFree(ptr)
        summary(f"  bare hipMalloc + hipFree of {gb} GB", t_new)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        {"plans": plans, "bare": bare}[sys.argv[1]]()
        sys.exit(0)
    for title, leg, env in (("library as it is (default: large blocks straight back to the runtime)", "plans", {"RTD_POOL_BYTES": "0"}),
                            ("library, opted in (rtd_pool_set_limit / RTD_POOL_BYTES = 36 GB: large blocks pooled)", "plans", {"RTD_POOL_BYTES": str(36 << 30)}),
                            ("runtime alone", "bare", {})):
        print(title, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), leg], env={**os.environ, **env}, check=True)
