#!/usr/bin/env python3
"""On the GPU box: every call form of the returned closures -- tau scalar / 0-d / list / array, phi likewise, antiderivative,
return_Fourier_error, return_tau_arr -- on random cases: the drop-in's shapes and values against the oracle's (whose call forms
are the reference's, except for two quirks of the reference recorded in DESIGN.md section 6).  Usage: python tools/fuzz_closures.py [ncases]"""
import itertools, os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd
from oracle import disort_oracle as O
import test_gpu_random_parity as T
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0


def shapes(x):
    return tuple(np.shape(y) for y in x) if isinstance(x, tuple) else np.shape(x)


def close(a, b, sc):
    if isinstance(a, tuple):
        return all(close(x, y, sc) for x, y in zip(a, b))
    return np.max(np.abs(np.asarray(a, float) - np.asarray(b, float)), initial=0.0) <= 1e-7 * sc


for seed in range(n):
    kw = T.make_case(seed) if seed % 3 else T.make_case_many_streams(seed)
    if np.any(np.asarray(kw["omega_arr"]) > 1 - 1e-5):
        continue
    g, o = pydisort_amd.pydisort(**kw), O.pydisort(**kw)
    tL = float(np.atleast_1d(kw["tau_arr"])[-1])
    sc = max(float(np.max(np.abs(o[3](np.array([0.0, tL]))))), 1e-300) * max(1.0, tL)
    taus = [0.3 * tL, np.array([0.3 * tL]), np.array([0.0, 0.5 * tL, tL]), np.float64(0.1 * tL), [0.2 * tL, 0.4 * tL], np.array(0.7 * tL)]
    phis = [1.0, np.array([1.0]), np.array([0.0, 2.0]), [0.5, 1.5, 2.5], np.float64(3.0)]
    if len(g) > 4:
        for tau, phi, anti, ferr, rta in itertools.product(taus, phis, (False, True), (False, True), (False, True)):
            try:
                a = g[4](tau, phi, anti, ferr, rta)
            except Exception as e:  # noqa: BLE001
                a = ("EXC " + type(e).__name__,)
            b = o[4](tau, phi, anti, ferr, rta)
            if shapes(a) != shapes(b) or not close(a, b, sc):
                bad += 1
                print("u", seed, type(tau).__name__, np.shape(tau), type(phi).__name__, np.shape(phi), anti, ferr, rta, shapes(a), shapes(b), flush=True)
    for tau, anti, rta in itertools.product(taus, (False, True), (False, True)):
        for idx, name in ((3, "u0"), (1, "flux_up"), (2, "flux_down")):
            try:
                a = g[idx](tau, anti, rta)
            except Exception as e:  # noqa: BLE001
                a = ("EXC " + type(e).__name__,)
            b = o[idx](tau, anti, rta)
            if shapes(a) != shapes(b) or not close(a, b, sc):
                bad += 1
                print(name, seed, type(tau).__name__, np.shape(tau), anti, rta, shapes(a), shapes(b), flush=True)
    g[1].__self__.plan.close()
print(f"{n} cases, {bad} findings")
