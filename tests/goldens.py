"""Loader for the golden vectors captured from the reference (tests/golden/ref/*.npz).

Format written by tests/golden/make_reference_goldens.py: flat keys
``c<i>.in.<arg>``, ``c<i>.bdrf<k>.{scalar|tab|tab0}``, ``c<i>.ev<j>.{name,arg<n>,kw.<k>,out<n>,nout}``.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "golden", "ref")
STAMNES_DIR = os.path.join(HERE, "golden", "stamnes")
_INT_ARGS = ("NQuad", "NLeg", "NFourier", "use_banded_solver_NLayers")
_BOOL_ARGS = ("only_flux", "NT_cor", "autograd_compatible")


class TabulatedBDRF:
    """BDRF Fourier mode replayed from captured tables: q(mu_i, mu_j) and q(mu_i, mu0)."""

    def __init__(self, tab, tab0):
        self.tab, self.tab0 = np.asarray(tab), np.asarray(tab0)

    def __call__(self, mu, neg_mup):
        neg_mup = np.atleast_1d(neg_mup)
        if len(neg_mup) == 1 and self.tab.shape[1] != 1:
            return self.tab0[:, None]
        return self.tab


def list_ids():
    return sorted(f[:-4] for f in os.listdir(REF_DIR) if f.endswith(".npz"))


def load(test_id):
    """-> list of calls; each call = dict(kwargs=..., mu_arr=..., evals=[dict(name,args,kwargs,out)])."""
    z = np.load(os.path.join(REF_DIR, test_id + ".npz"), allow_pickle=False)
    keys = set(z.files)
    calls = []
    for ci in range(int(z["ncalls"])):
        p = f"c{ci}"
        kw = {}
        for k in keys:
            if k.startswith(p + ".in."):
                name = k[len(p) + 4:]
                if f"{p}.none.{name}" in keys:
                    kw[name] = None
                elif name in _INT_ARGS:
                    kw[name] = int(z[k])
                elif name in _BOOL_ARGS:
                    kw[name] = bool(z[k])
                else:
                    v = z[k]
                    kw[name] = float(v) if v.ndim == 0 else np.array(v)
        modes = []
        for mi in range(int(z[p + ".nbdrf"])):
            if f"{p}.bdrf{mi}.scalar" in keys:
                modes.append(float(z[f"{p}.bdrf{mi}.scalar"]))
            else:
                modes.append(TabulatedBDRF(z[f"{p}.bdrf{mi}.tab"], z[f"{p}.bdrf{mi}.tab0"]))
        kw["BDRF_Fourier_modes"] = modes
        evals = []
        for ei in range(int(z[p + ".nevals"])):
            q = f"{p}.ev{ei}"
            args = []
            for ai in range(int(z[q + ".nargs"])):
                a = z[f"{q}.arg{ai}"]
                args.append(a[()] if a.ndim == 0 else np.array(a))
            ekw = {k[len(q) + 4:]: z[k][()] for k in keys if k.startswith(q + ".kw.")}
            nout = int(z[q + ".nout"])
            out = z[q + ".out0"] if nout == 0 else tuple(z[f"{q}.out{i}"] for i in range(nout))
            evals.append(dict(name=str(z[q + ".name"]), args=args, kwargs=ekw, out=out))
        calls.append(dict(kwargs=kw, mu_arr=z[p + ".mu_arr"], evals=evals))
    return calls


def stamnes(test_id):
    return np.load(os.path.join(STAMNES_DIR, test_id + "_test.npz"))


def nt_is_active(kw):
    """Condition of pydisort.py:375 for the Nakajima-Tanaka corrections being applied."""
    f = np.atleast_1d(kw.get("f_arr", 0))
    leg = np.atleast_2d(kw["Leg_coeffs_all"])
    nleg = kw["NLeg"] if kw.get("NLeg") is not None else kw["NQuad"]
    return bool(kw.get("NT_cor")) and not kw.get("only_flux") and kw["I0"] > 0 and np.any(f > 0) \
        and nleg < leg.shape[1] and np.any(np.atleast_1d(kw["omega_arr"]) > 0)


def uses_antiderivative(ev):
    a = ev["args"]
    if ev["name"] == "u":
        return (len(a) > 2 and bool(a[2])) or bool(ev["kwargs"].get("is_antiderivative_wrt_tau", False))
    return (len(a) > 1 and bool(a[1])) or bool(ev["kwargs"].get("is_antiderivative_wrt_tau", False))


def max_rel_err(got, want):
    """max |got-want| / max(|want|) over the array (scale-relative), plus pointwise rel on significant points."""
    got, want = np.asarray(got, float), np.asarray(want, float)
    scale = np.max(np.abs(want)) if want.size else 0.0
    if scale == 0:
        return float(np.max(np.abs(got), initial=0.0)), 0.0
    diff = np.abs(got - want)
    sig = np.abs(want) > 1e-8 * scale
    pw = float(np.max(diff[sig] / np.abs(want[sig]), initial=0.0))
    return float(diff.max() / scale), pw
