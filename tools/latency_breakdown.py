#!/usr/bin/env python3
"""Where the time of a one-column pydisort() call goes (Test Problem 9c by default; any golden test id as argument): wall-clock spans of the host stages and of every
C-ABI call, median over 200 calls (run on the GPU box)."""
import os, sys, time, warnings, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import goldens, pydisort_amd
import importlib
from pydisort_amd import _engine, _prepare, _lib
P = importlib.import_module("pydisort_amd.pydisort")
warnings.simplefilter("ignore")
TID = sys.argv[1] if len(sys.argv) > 1 else "9c"  # a golden test id: 9c (6 layers, 8 streams), 5a (Cloud C.1, 48 streams, NT corrections), ...
kw = goldens.load(TID)[0]["kwargs"]
spans = collections.defaultdict(list)


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            spans[name].append(time.perf_counter() - t0)
    return w


lib = _lib.load()
for name in ("rtd_plan_create_retained", "rtd_plan_set_quadrature", "rtd_plan_set_columns", "rtd_plan_solve", "rtd_plan_evaluate",
             "rtd_plan_destroy"):
    setattr(lib, name, timed("C " + name, getattr(lib, name)))
P.prepare_columns = timed("py prepare_columns", P.prepare_columns)
P._plan_for = timed("py _plan_for (incl. C set_columns / create)", P._plan_for)
P._tabulate_bdrf = timed("py _tabulate_bdrf", P._tabulate_bdrf)
_engine.Plan.evaluate = timed("py Plan.evaluate (incl. C evaluate)", _engine.Plan.evaluate)
for _ in range(20):
    res = pydisort_amd.pydisort(**kw); res[1](0.5)
spans.clear()
tot, ev = [], []
for _ in range(200):
    t0 = time.perf_counter()
    res = pydisort_amd.pydisort(**kw)
    t1 = time.perf_counter()
    res[1](0.5)
    t2 = time.perf_counter()
    tot.append(t1 - t0); ev.append(t2 - t1)
med = lambda v: 1e6 * sorted(v)[len(v) // 2]
print(f"pydisort() {med(tot):7.1f} us   first flux_up() {med(ev):7.1f} us   (medians of 200 calls)")
for k, v in sorted(spans.items()):
    print(f"  {k:48s} {med(v):7.1f} us  x {len(v) / 200:.1f} per call")
