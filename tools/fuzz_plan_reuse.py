#!/usr/bin/env python3
"""Random sequences of calls on ONE long-lived plan (the serving pattern: new inputs into an existing plan, run, fetch,
evaluate elsewhere, new evaluation points, run again ...), windowed and pipelined plans included; after every step that
returns results they must equal, bit for bit, those of a fresh plan given the same inputs.  Hunts stale state (cached
tables, stream forks, hand-off slots, status words).  Usage: python tools/fuzz_plan_reuse.py [nplans]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd
from pydisort_amd import synthetic
warnings.simplefilter("ignore")
nplans = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
t0 = time.time()


def make_inputs(rng, C, L, NQuad, with_thermal, with_bc):
    g = rng.uniform(0.0, 0.85, (C, L))
    cfg = dict(tau_arr=np.cumsum(10.0 ** rng.uniform(-2, 0.6, (C, L)), axis=1), omega_arr=rng.uniform(0.05, 0.99, (C, L)), NQuad=NQuad,
               Leg_coeffs_all=g[:, :, None] ** np.arange(NQuad + 1)[None, None, :], mu0=rng.uniform(0.15, 1.0, C), I0=rng.uniform(0.5, 3.0, C),
               phi0=rng.uniform(0, 6.0, C), f_arr=g**NQuad)
    if with_thermal:
        cfg["s_poly_coeffs"] = rng.uniform(0, 1, (C, L, 2))
    if with_bc:
        cfg["b_neg"] = float(rng.uniform(0.05, 1))
    return cfg


def points(rng, cfg, interfaces):
    C = cfg["tau_arr"].shape[0]
    if interfaces:
        tau = np.concatenate((np.zeros((C, 1)), cfg["tau_arr"]), axis=1)
    else:
        tau = np.sort(rng.uniform(0, 1, (C, int(rng.integers(1, 5)))), axis=1) * cfg["tau_arr"][:, -1:]
    return tau, rng.uniform(0, 6.28, int(rng.integers(1, 4)))


for pl in range(nplans):
    rng = np.random.default_rng([99, pl])
    NQuad = int(rng.choice([4, 8, 16, 32, 32, 32, 64]))
    C = int(rng.integers(2, 48 if NQuad <= 32 else 10))
    L = int(rng.integers(1, 6))
    M = int(rng.integers(1, min(NQuad, 6) + 1))
    wc = int(rng.choice([0, 1, max(1, C // 3), max(1, C // 2), C]))
    raw = bool(rng.random() < 0.5)
    th, bc = bool(rng.random() < 0.4), bool(rng.random() < 0.4)
    # round 6: the long-lived plan in a retained form (full / lean: evaluations after a run read retained state, the lean form
    # re-runs the eigen stage for the chunks the points touch); the fresh plan it is compared with never is
    retain = (False, "full", "lean")[int(rng.integers(0, 3))]
    tag = f"plan {pl}: NQuad {NQuad} C {C} L {L} M {M} work_columns {wc} raw {raw} retain {retain}"
    try:
        cfg = make_inputs(rng, C, L, NQuad, th, bc)
        cfg["NFourier"] = M
        _, sol = pydisort_amd.pydisort_batch(work_columns=wc, device_prepare=raw, retain=retain, _defer_solve=True, **cfg)
        plan = sol.plan
        tau, phi = points(rng, cfg, True)
        have_points = False
        for step in range(int(rng.integers(4, 12))):
            op = rng.choice(["new_inputs", "run_fetch", "run+fetch", "evaluate", "new_points", "run_only", "run_fetch"])
            if op == "new_inputs":
                cfg = make_inputs(rng, C, L, NQuad, th, bc)
                cfg["NFourier"] = M
                _, fresh_for_prep = pydisort_amd.pydisort_batch(work_columns=wc, device_prepare=raw, _defer_solve=True, **cfg)
                if raw:
                    plan.set_columns_raw(fresh_for_prep.plan.prep["raw"])
                else:
                    plan.set_columns(fresh_for_prep.plan.prep)
                fresh_for_prep.plan.close()
                sol.prep = plan.prep
                tau, phi = points(rng, cfg, True)  # (the old points may lie outside the new atmospheres)
                have_points = False
                continue
            if op == "new_points":
                tau, phi = points(rng, cfg, bool(rng.random() < 0.5))
                plan.set_eval_points(tau, phi)
                have_points = True
                continue
            if not have_points:
                plan.set_eval_points(tau, phi)
                have_points = True
            if op == "run_only":
                plan.run()
                continue
            if op == "evaluate":
                if not getattr(plan, "solved", False):
                    plan.solve()
                t2, p2 = points(rng, cfg, False)
                anti = bool(rng.random() < 0.3)
                got = plan.evaluate(t2, p2, anti)
                have_points = False  # evaluate replaces the run-path points
                ref_t, ref_p = t2, p2
                kind = "evaluate"
            else:
                if op == "run_fetch":
                    got = plan.run_fetch()
                else:
                    plan.run()
                    got = plan.fetch()
                ref_t, ref_p = tau, phi
                kind = op
            _, fs = pydisort_amd.pydisort_batch(work_columns=wc, device_prepare=raw, _defer_solve=True, **cfg)
            if kind != "evaluate":  # the same entry point on the fresh plan (run-path points at the interfaces are evaluated
                fs.plan.set_eval_points(ref_t, ref_p)  # inside the boundary-condition kernel: 1e-11 from the closures' kernel)
                want = fs.plan.run_fetch()
            else:
                fs.plan.solve()
                want = fs.plan.evaluate(ref_t, ref_p, anti)
            if True:
                for k in ("u", "u0", "flux_up", "flux_down_diffuse"):
                    if not np.array_equal(got[k], want[k]):
                        d = np.max(np.abs(got[k] - want[k])) / max(np.max(np.abs(want[k])), 1e-300)
                        if d > 1e-12:
                            bad += 1
                            print(tag, f"step {step} {kind}: {k} differs from a fresh plan by {d:.2e}", flush=True)
                        break
            fs.plan.close()
        plan.close()
    except Exception as e:
        bad += 1
        print(tag, "EXCEPTION", type(e).__name__, str(e)[:200], flush=True)
print(f"{nplans} plans, {bad} findings, {time.time() - t0:.0f} s")
