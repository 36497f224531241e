#!/usr/bin/env python3
"""Capture golden vectors from the reference PythonicDISORT (THIS CONTAINER ONLY).

Runs every test function of /root/reference/pydisotest/*_test.py with a recording
wrapper around ``PythonicDISORT.pydisort``.  For each pydisort call we store

* every input argument (BDRF callables are replaced by their values on the
  quadrature grid: q(mu_i, mu_j) [N,N] and q(mu_i, mu0) [N]), and
* every evaluation the test performed on the returned closures
  (closure name, its arguments, and the float64 result).

Output: tests/golden/ref/<test id>.npz  (flat keys, see tests/goldens.py for the
loader) and a copy of the Stamnes DISORT 4.0.99 result files (data, MIT licensed)
under tests/golden/stamnes/.

Nothing of the reference's *source* is stored -- only inputs and outputs.

Usage:  PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_reference_goldens.py
"""
import importlib.util
import inspect
import os
import shutil
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref")
MAX_EVALS_PER_CLOSURE = 100000  # every evaluation a test performs (round 5; rounds 1-4 kept the first ten per closure)

sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "pydisotest"))
import PythonicDISORT  # noqa: E402
from PythonicDISORT import subroutines as ref_sub  # noqa: E402

_real_pydisort = PythonicDISORT.pydisort
_sig = inspect.signature(_real_pydisort)

_calls = []  # records of the current test


class _Tabulated:
    """Callable standing in for a BDRF Fourier mode; returns captured tables."""

    def __init__(self, f, mu_pos, mu0, beam):
        self.mu_pos = mu_pos
        self.mu0 = mu0
        self.tab = np.asarray(f(mu_pos, mu_pos), dtype=float)
        self.tab0 = (
            np.asarray(f(mu_pos, np.array([mu0])), dtype=float)[:, 0]
            if beam
            else np.zeros(len(mu_pos))
        )

    def __call__(self, mu, neg_mup):
        mu = np.atleast_1d(mu)
        neg_mup = np.atleast_1d(neg_mup)
        if len(neg_mup) == len(self.mu_pos) and np.allclose(neg_mup, self.mu_pos):
            return self.tab
        if len(neg_mup) == 1 and np.isclose(neg_mup[0], self.mu0):
            return self.tab0[:, None]
        raise RuntimeError("unexpected BDRF evaluation point")


def _wrap_closure(rec, name, fn):
    def wrapped(*args, **kwargs):
        out = fn(*args, **kwargs)
        evs = rec["evals"]
        if sum(1 for e in evs if e["name"] == name) < MAX_EVALS_PER_CLOSURE:
            evs.append(dict(name=name, args=args, kwargs=dict(kwargs), out=out))
        return out

    # subroutines.interpolate inspects __code__.co_argcount; keep the original reachable
    wrapped.__wrapped__ = fn
    return wrapped


def recording_pydisort(*args, **kwargs):
    ba = _sig.bind(*args, **kwargs)
    ba.apply_defaults()
    a = dict(ba.arguments)
    NQuad = int(a["NQuad"])
    N = NQuad // 2
    mu_pos, _ = ref_sub.Gauss_Legendre_quad(N)
    beam = a["I0"] > 0
    modes = []
    for f in a["BDRF_Fourier_modes"]:
        if np.isscalar(f):
            modes.append(f)
        else:
            modes.append(_Tabulated(f, mu_pos, a["mu0"], beam))
    a["BDRF_Fourier_modes"] = modes
    # the reference mutates Leg_coeffs_all[:,0] in place; keep a pristine copy as the input
    inputs = {k: (np.array(v, dtype=float, copy=True) if k not in
                  ("BDRF_Fourier_modes", "NLeg", "NFourier", "NQuad", "only_flux", "NT_cor",
                   "use_banded_solver_NLayers", "autograd_compatible") else v)
              for k, v in a.items()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = _real_pydisort(**a)
    rec = dict(inputs=inputs, modes=modes, evals=[], mu_arr=np.array(res[0]))
    _calls.append(rec)
    names = ["flux_up", "flux_down", "u0", "u"][: len(res) - 1]
    return (res[0],) + tuple(_wrap_closure(rec, n, f) for n, f in zip(names, res[1:]))


def _flatten_out(prefix, out, store):
    if isinstance(out, tuple):
        store[prefix + ".nout"] = np.array(len(out))
        for i, o in enumerate(out):
            store[f"{prefix}.out{i}"] = np.asarray(o, dtype=float)
    else:
        store[prefix + ".nout"] = np.array(0)  # 0 => not a tuple
        store[prefix + ".out0"] = np.asarray(out, dtype=float)


def dump(test_id):
    store = {"ncalls": np.array(len(_calls))}
    for ci, rec in enumerate(_calls):
        p = f"c{ci}"
        for k, v in rec["inputs"].items():
            if k == "BDRF_Fourier_modes":
                continue
            if v is None:
                store[f"{p}.in.{k}"] = np.array(np.nan)
                store[f"{p}.none.{k}"] = np.array(1)
            else:
                store[f"{p}.in.{k}"] = np.asarray(v)
        store[f"{p}.nbdrf"] = np.array(len(rec["modes"]))
        for mi, f in enumerate(rec["modes"]):
            if np.isscalar(f):
                store[f"{p}.bdrf{mi}.scalar"] = np.array(float(f))
            else:
                store[f"{p}.bdrf{mi}.tab"] = f.tab
                store[f"{p}.bdrf{mi}.tab0"] = f.tab0
        store[f"{p}.mu_arr"] = rec["mu_arr"]
        store[f"{p}.nevals"] = np.array(len(rec["evals"]))
        for ei, ev in enumerate(rec["evals"]):
            q = f"{p}.ev{ei}"
            store[q + ".name"] = np.array(ev["name"])
            store[q + ".nargs"] = np.array(len(ev["args"]))
            for ai, arg in enumerate(ev["args"]):
                store[f"{q}.arg{ai}"] = np.asarray(arg)
            for k, v in ev["kwargs"].items():
                store[f"{q}.kw.{k}"] = np.asarray(v)
            _flatten_out(q, ev["out"], store)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, test_id + ".npz"), **store)


def main():
    PythonicDISORT.pydisort = recording_pydisort
    os.chdir(os.path.join(REF, "pydisotest"))
    np.random.seed(11)  # test_11a draws unseeded random tau points; pin them
    files = sorted(f for f in os.listdir(".") if f.endswith("_test.py"))
    only = set(sys.argv[1:])
    for fn in files:
        spec = importlib.util.spec_from_file_location("ref_" + fn[:-3].replace(".", "_"), fn)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for name, func in sorted(vars(mod).items()):
            if not (name.startswith("test_") and callable(func)):
                continue
            test_id = name[len("test_"):]
            if only and test_id not in only:
                continue
            _calls.clear()
            # silence the reference's prints
            devnull = open(os.devnull, "w")
            old = sys.stdout
            sys.stdout = devnull
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    func()
            finally:
                sys.stdout = old
                devnull.close()
            dump(test_id)
            print(f"{test_id}: {len(_calls)} pydisort calls, "
                  f"{sum(len(c['evals']) for c in _calls)} evaluations")
    # Stamnes result files (pure data)
    dst = os.path.join(HERE, "stamnes")
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir("Stamnes_results"):
        shutil.copyfile(os.path.join("Stamnes_results", f), os.path.join(dst, f))


if __name__ == "__main__":
    main()
