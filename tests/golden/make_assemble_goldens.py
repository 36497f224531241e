#!/usr/bin/env python3
"""Positional arguments of the reference's INNER boundary, captured from the reference itself (THIS CONTAINER ONLY).

For a selection of the reference-captured pydisort calls (tests/golden/ref/<id>.npz) the reference's ``pydisort`` is run
again with a recorder in place of ``_assemble_intensity_and_fluxes`` (the name ``pydisort.py:3`` imports): the recorder
stores the 34 positional arguments exactly as ``pydisort.py:381-405`` / ``:701-725`` hand them down -- BDRF callables as
their tables on the quadrature grid, q(mu_i, mu_j) and q(mu_i, mu0) -- calls the real function, evaluates the returned
callables at fixed points and stores those results.  Output: tests/golden/assemble/<id>.npz (data only).

Usage:  PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_assemble_goldens.py
"""
import inspect
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = ["/root/reference/src", os.path.dirname(HERE)]
import PythonicDISORT  # noqa: E402
import goldens  # noqa: E402

MOD = sys.modules["PythonicDISORT.pydisort"]
REAL = MOD._assemble_intensity_and_fluxes
NAMES = list(inspect.signature(REAL).parameters)
assert len(NAMES) == 34, NAMES
CASES = ["1a", "2c", "3a", "4c", "5b", "6d", "7a", "8a", "8ARTS_B", "9c", "11a", "Ia"]
PHI = np.array([0.0, np.pi / 2, 2.5])
_rec = []


def recorder(*args):
    assert len(args) == 34
    a = dict(zip(NAMES, args))
    tau_arr = np.asarray(a["tau_arr"], float)
    mids = 0.5 * (np.concatenate(([0.0], tau_arr[:-1])) + tau_arr)
    tau_pts = np.sort(np.concatenate(([0.0], tau_arr, mids)))
    res = REAL(*args)
    out = {"flux_up": res[0](tau_pts), "u0": res[2](tau_pts)}
    out["flux_down_diffuse"], out["flux_down_direct"] = res[1](tau_pts)
    if len(res) > 3:
        out["u"] = res[3](tau_pts, PHI)
    _rec.append((a, tau_pts, out))
    return res


def dump(test_id):
    store = {"ncalls": np.array(len(_rec)), "names": np.array(NAMES), "phi": PHI}
    for ci, (a, tau_pts, out) in enumerate(_rec):
        p = f"c{ci}"
        mu, mu0 = np.asarray(a["mu_arr_pos"], float), a["mu0"]
        for k, v in a.items():
            if k == "BDRF_Fourier_modes":
                store[f"{p}.nbdrf"] = np.array(len(v))
                for mi, f in enumerate(v):
                    if np.isscalar(f):
                        store[f"{p}.bdrf{mi}.scalar"] = np.array(float(f))
                    else:
                        store[f"{p}.bdrf{mi}.tab"] = np.asarray(f(mu, mu), float)
                        store[f"{p}.bdrf{mi}.tab0"] = (np.asarray(f(mu, np.array([mu0])), float)[:, 0]
                                                       if a["there_is_beam_source"] else np.zeros(len(mu)))
            elif v is None:
                store[f"{p}.none.{k}"] = np.array(1)
            else:
                store[f"{p}.arg.{k}"] = np.asarray(v)
        store[f"{p}.tau_pts"] = tau_pts
        for k, v in out.items():
            store[f"{p}.out.{k}"] = np.asarray(v, float)
    os.makedirs(os.path.join(HERE, "assemble"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "assemble", test_id + ".npz"), **store)
    print(test_id, len(_rec), "call(s)", flush=True)


if __name__ == "__main__":
    MOD._assemble_intensity_and_fluxes = recorder
    for test_id in CASES:
        _rec.clear()
        for call in goldens.load(test_id)[:3]:  # (8ARTS_B and the like make many calls: three are enough)
            kw = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in call["kwargs"].items()}
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                PythonicDISORT.pydisort(**kw)
        dump(test_id)
