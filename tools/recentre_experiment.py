"""NumPy experiment behind csrc/rtd_dd.h (round 5): thermal source polynomials re-centred per layer (exact rational shift) in the
oracle -> 8ARTS_A against the 40-digit truth: 3.0e-5 -> 2.0e-6 pointwise.  The library does this on upload since; the oracle stays the
reference's algorithm (absolute form)."""
import os, sys
from fractions import Fraction
from math import comb
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (tools/ -> repo root)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import goldens
from oracle import disort_oracle as O

def local_coeffs(s_row, top):
    n = len(s_row); a = [Fraction(float(x)) for x in s_row]; t = Fraction(float(top))
    return np.array([float(sum(a[j] * comb(j, i) * t ** (j - i) for j in range(i, n))) for i in range(n)])

orig_prepare, orig_v = O.prepare, O.mathscr_v
def prepare(*a, **k):
    p = orig_prepare(*a, **k)
    if p["Ns"] > 0:
        assert np.all(p["scale_tau"] == 1.0)
        s_abs = p["s_s"]; ts0 = p["tau_s0"]
        p["s_s"] = np.stack([local_coeffs(s_abs[l], ts0[l]) for l in range(len(s_abs))])
    return p
def mathscr_v(p, G0, K0, zvec, taus, l, rows=slice(None), antider=False):
    assert not antider
    ts0 = p["tau_s0"]
    return orig_v(p, G0, K0, zvec, [t - ts0[ll] for t, ll in zip(taus, l)], l, rows, antider)

z = np.load(os.path.join(ROOT, "tests", "golden", "hp", "golden_8ARTS_A.npz"))
for label, patch in (("absolute (reference's form)", False), ("re-centred per layer", True)):
    O.prepare, O.mathscr_v = (prepare, mathscr_v) if patch else (orig_prepare, orig_v)
    worst = worst_pw = 0.0
    for ci, call in enumerate(goldens.load("8ARTS_A")):
        if f"c{ci}.u" not in z.files: continue
        ev = next(e for e in call["evals"] if e["name"] == "u" and not e["kwargs"] and len(e["args"]) == 2)
        got = O.pydisort(**call["kwargs"])[4](*ev["args"])
        a, b = goldens.max_rel_err(got, z[f"c{ci}.u"])
        worst, worst_pw = max(worst, a), max(worst_pw, b)
    print(label, "scale-rel", worst, "pointwise", worst_pw)
