#!/usr/bin/env python3
"""GPU-box diagnostic: replay the reference goldens through the HIP path and print the errors."""
import os
import sys
import time
import traceback
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import goldens  # noqa: E402
import pydisort_amd  # noqa: E402

warnings.simplefilter("ignore")
ids = sys.argv[1:] or goldens.list_ids()
for tid in ids:
    try:
        worst = 0.0
        t0 = time.time()
        for call in goldens.load(tid):
            res = pydisort_amd.pydisort(**call["kwargs"])
            fns = dict(zip(["flux_up", "flux_down", "u0", "u"], res[1:]))
            scale = max(max(np.max(np.abs(o), initial=0.0) for o in
                            (ev["out"] if isinstance(ev["out"], tuple) else (ev["out"],))) for ev in call["evals"])
            for ev in call["evals"]:
                got = fns[ev["name"]](*ev["args"], **ev["kwargs"])
                gots = got if isinstance(got, tuple) else (got,)
                wants = ev["out"] if isinstance(ev["out"], tuple) else (ev["out"],)
                for g, w in zip(gots, wants):
                    e = float(np.max(np.abs(np.asarray(g) - w), initial=0.0)) / scale
                    if not np.isfinite(e):
                        e = float("inf")
                    worst = max(worst, e)
        print(f"{tid:14s} err={worst:.2e}  ({time.time() - t0:.2f}s)", flush=True)
    except Exception:
        print(f"{tid:14s} EXCEPTION", flush=True)
        traceback.print_exc()
