#!/usr/bin/env python3
"""Calibration of bench.py's cpu_baseline ("ratio restatement / reference on identical hardware").

bench.py times the oracle (oracle/disort_oracle.py, the NumPy/SciPy restatement of the reference's algorithm) on the GPU
box's host cores, because the reference itself cannot travel there.  This script runs in the BUILD container, where
/root/reference is present: the oracle and the reference (PythonicDISORT.pydisort) solve the same seeded cfg4 columns on
the same core, one BLAS thread, interleaved column by column; the ratio of their rates goes to
profiles/archive/r02_cpu_calibration.json, which bench.py copies into its cpu_baseline object ("calibration").

Usage:  PYTHONDONTWRITEBYTECODE=1 python3 tools/calibrate_cpu_baseline.py [columns]
"""
import json
import os
import sys
import time
import warnings

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = ["/root/reference/src", ROOT, os.path.join(ROOT, "pythonic-disort_amd")]
from threadpoolctl import threadpool_limits  # noqa: E402
import PythonicDISORT  # noqa: E402
from oracle import disort_oracle as O  # noqa: E402
from pydisort_amd import synthetic  # noqa: E402


def solve(fn, kw, tau, phi):
    res = fn(**kw)
    u = res[4](tau, phi)
    res[1](tau), res[2](tau)
    return u


def main(ncols):
    phi = np.array([0.0, np.pi / 2, np.pi])
    t_ref = t_orc = 0.0
    worst = 0.0
    warnings.simplefilter("ignore")
    with threadpool_limits(1):
        for i in range(-2, ncols):  # two untimed warm-up columns
            cfg = synthetic.cfg4_columns(1, first=20_000 + max(i, 0))
            kw = synthetic.column_kwargs(cfg, 0)
            tau = np.concatenate(([0.0], cfg["tau_arr"][0]))
            t0 = time.perf_counter()
            a = solve(PythonicDISORT.pydisort, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()}, tau, phi)
            t1 = time.perf_counter()
            b = solve(O.pydisort, kw, tau, phi)
            t2 = time.perf_counter()
            if i >= 0:
                t_ref += t1 - t0
                t_orc += t2 - t1
                worst = max(worst, float(np.max(np.abs(a - b)) / np.max(np.abs(a))))
    out = dict(r=t_ref / t_orc, reference_columns_per_s=ncols / t_ref, oracle_columns_per_s=ncols / t_orc, columns=ncols,
               max_rel_output_difference=worst,
               workload="cfg4 columns 20000.. (L=20, NQuad=32, 32 Fourier modes, u at 21 tau x 3 phi + fluxes), 1 BLAS thread, "
                        "same core, reference and oracle interleaved",
               host=dict(cpus=os.cpu_count(), numpy=np.__version__),
               meaning="r = oracle rate / reference rate: cpu_baseline.value / r is what the reference would do on the GPU box's cores")
    path = os.path.join(ROOT, "profiles", "archive", "r02_cpu_calibration.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
