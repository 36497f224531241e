#!/usr/bin/env python3
"""On the GPU box: the HIP path against the 40-digit solution of tools/hp_truth_m0.py (tests/golden/hp_truth_m0.npz)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd"), os.path.join(ROOT, "tools")]
import pydisort_amd
import importlib.util
spec = importlib.util.spec_from_file_location("hp", os.path.join(ROOT, "tools", "hp_cases.py"))
hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)
Z = np.load(os.path.join(ROOT, "tests", "golden", "hp_truth_m0.npz"))
for name, kw in (("benign", hp.benign_case()), ("harsh", hp.harsh_case())):
    _, fu, fd, u0 = pydisort_amd.pydisort(**kw)
    tau = np.concatenate(([0.0], kw["tau_arr"]))
    got = u0(tau)
    truth, orc = Z[name], Z[name + "_oracle"]
    s = np.max(np.abs(truth))
    print(f"{name:7s} HIP vs truth {np.max(np.abs(got - truth)) / s:.2e}   oracle vs truth {np.max(np.abs(orc - truth)) / s:.2e}   HIP vs oracle {np.max(np.abs(got - orc)) / s:.2e}")
kw = hp.intensity_case()
_, fu, fd, u0, u = pydisort_amd.pydisort(**kw)
tau = np.concatenate(([0.0], kw["tau_arr"]))
got, truth, orc = u(tau, hp.PHI), Z["intensity"], Z["intensity_oracle"]
s = np.max(np.abs(truth))
print(f"intensity HIP vs truth {np.max(np.abs(got - truth)) / s:.2e}   oracle vs truth {np.max(np.abs(orc - truth)) / s:.2e}")
