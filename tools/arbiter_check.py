#!/usr/bin/env python3
"""Is the arbiter itself right?  The 40-digit solutions of tools/hp_truth_case.py decide every HIP-vs-oracle disagreement
above 1e-6 -- but the generator was written by the builder and checked against the builder's oracle.  This script pins it to
something the builder did not write: the outputs of the REFERENCE itself (PythonicDISORT run in the build container, captured
in tests/golden/ref/*.npz by tests/golden/make_reference_goldens.py).  For every captured pydisort() call of every golden case
the 40-digit solution is computed at the very points the reference was evaluated at and compared with what the reference
returned:

  * flux_up(tau)            -- every call (independent of the Nakajima-Tanaka corrections);
  * u(tau, phi)             -- every call made with NT_cor off (the truth machinery solves the uncorrected problem).

Well-conditioned calls must agree to ~1e-12 (two float64-vs-40-digit roundoff levels); where they do not, the case is one of
the near-conservative / 8ARTS ones whose float64 conditioning the docs discuss, and the number is printed as it is.

Writes profiles/archive/r04_arbiter_vs_reference.json.  Usage (build container, ~1 h on 6 processes):
    python3 tools/arbiter_check.py [--budget-seconds 3600] [case ...]
"""
import json
import multiprocessing
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import goldens  # noqa: E402
import hp_truth_case as H  # noqa: E402

OUT = os.path.join(ROOT, "profiles", "archive", "r04_arbiter_vs_reference.json")


def _one(job):
    kw, tau_u, phi, tau_f = job
    kw = {k: v for k, v in kw.items() if k != "autograd_compatible"}
    only_flux = bool(kw.get("only_flux", False))
    t0 = time.time()
    # one truth evaluation at the union of the points (u points first, then the flux points)
    tau_all = np.concatenate([np.atleast_1d(tau_u) if tau_u is not None else np.zeros(0), np.atleast_1d(tau_f)])
    u, u0, fup = H.truth(kw, tau_all, np.atleast_1d(phi) if phi is not None else np.array([0.0]), parallel=False)
    nu = 0 if tau_u is None else len(np.atleast_1d(tau_u))
    return (None if (u is None or only_flux or nu == 0) else u[:, :nu, :]), fup[nu:], time.time() - t0


def cost(call):
    kw = call["kwargs"]
    nq, L = kw["NQuad"], np.size(kw["tau_arr"])
    M = 1 if kw.get("only_flux") else (kw.get("NFourier") or nq)
    return M * L * nq**3


def main():
    budget = 3600.0
    names = []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--budget-seconds":
            budget = float(args.pop(0))
        else:
            names.append(a)
    if not names:
        names = sorted(f[:-4] for f in os.listdir(os.path.join(ROOT, "tests", "golden", "ref")) if f.endswith(".npz"))
    jobs, meta = [], []
    for name in names:
        for ci, call in enumerate(goldens.load(name)):
            kw = call["kwargs"]
            ev_u = next((e for e in call["evals"] if e["name"] == "u" and not e["kwargs"] and len(e["args"]) == 2), None)
            ev_f = next((e for e in call["evals"] if e["name"] == "flux_up" and not e["kwargs"] and len(e["args"]) == 1), None)
            nt = bool(kw.get("NT_cor", False))
            if ev_f is None and (ev_u is None or nt):
                continue
            use_u = ev_u is not None and not nt and not kw.get("only_flux", False)
            tau_f = np.atleast_1d(ev_f["args"][0]) if ev_f is not None else np.array([0.0])
            jobs.append((kw, ev_u["args"][0] if use_u else None, ev_u["args"][1] if use_u else None, tau_f))
            meta.append((name, ci, ev_u["out"] if use_u else None, ev_f["out"] if ev_f is not None else None, cost(call), nt))
    order = np.argsort([m[4] for m in meta])  # cheap first: the budget cuts the expensive tail, not the breadth
    t0 = time.time()
    results = {}
    skipped = []
    with multiprocessing.Pool(int(os.environ.get("HP_WORKERS", "6"))) as pool:
        pending = []
        for i in order:
            pending.append((i, pool.apply_async(_one, (jobs[i],))))
        for i, r in pending:
            name, ci, ref_u, ref_f, _, nt = meta[i]
            left = budget - (time.time() - t0)
            try:
                u, fup, secs = r.get(timeout=max(left, 1.0))
            except multiprocessing.TimeoutError:
                skipped.append(f"{name}/c{ci}")
                continue
            rec = {"seconds": round(secs, 1), "NT_cor": nt}
            if ref_f is not None:
                ref_f = np.atleast_1d(np.asarray(ref_f, float))
                rec["flux_up_rel"] = float(np.max(np.abs(ref_f - fup)) / max(np.max(np.abs(fup)), 1e-300))
            if u is not None and ref_u is not None:
                ref_u = np.asarray(ref_u, float)
                u = u.reshape(ref_u.shape)
                scale = max(float(np.max(np.abs(u))), 1e-300)
                rec["u_scale_rel"] = float(np.max(np.abs(ref_u - u)) / scale)
                big = np.abs(u) > 1e-8 * scale
                if np.any(big):  # (an atmosphere without sources: the field is identically zero)
                    rec["u_pointwise_rel"] = float(np.max(np.abs(ref_u - u)[big] / np.abs(u)[big]))
            results.setdefault(name, {})[f"c{ci}"] = rec
            print(name, ci, rec, flush=True)
            _write(results, skipped, time.time() - t0)
        pool.terminate()
    _write(results, skipped, time.time() - t0)


def _write(results, skipped, secs):
    worst = {}
    for name, calls in results.items():
        for k in ("flux_up_rel", "u_scale_rel", "u_pointwise_rel"):
            v = [c[k] for c in calls.values() if k in c]
            if v:
                worst.setdefault(name, {})[k] = max(v)
    well = {n: w for n, w in worst.items() if max(w.get("u_scale_rel", 0.0), w.get("flux_up_rel", 0.0)) <= 1e-10}
    doc = {"what": "40-digit solutions of tools/hp_truth_case.py against the REFERENCE's own captured outputs (tests/golden/ref: PythonicDISORT "
                   "run in the build container) at the reference's evaluation points: the arbiter pinned to something the builder did not write",
           "tool": "tools/arbiter_check.py", "seconds": round(secs), "cases": len(results), "calls": sum(len(c) for c in results.values()),
           "cases_within_1e-10_of_the_reference": len(well),
           "max_over_those": {k: max((w.get(k, 0.0) for w in well.values()), default=None) for k in ("flux_up_rel", "u_scale_rel", "u_pointwise_rel")},
           "cases_beyond_1e-10": {n: w for n, w in worst.items() if n not in well},
           "not_finished_within_the_budget": skipped, "per_case_worst": worst, "per_call": results}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
