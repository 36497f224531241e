#!/usr/bin/env python3
"""Golden vectors for BASELINE.json's synthetic configs, computed by the REFERENCE (this container only).

For the first few columns of each synthetic config (pydisort_amd.synthetic) the reference
PythonicDISORT is called one column at a time and u, u0, flux_up, flux_down are stored at every layer
interface, mid-layer points and phi in {0, pi/2, pi, 2.5}.  Inputs are regenerated deterministically by
pydisort_amd.synthetic, so only outputs (and the evaluation points) are stored.

Usage:  PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_synthetic_goldens.py
"""
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = ["/root/reference/src", os.path.join(ROOT, "pythonic-disort_amd")]
import PythonicDISORT  # noqa: E402
from pydisort_amd import synthetic  # noqa: E402

PHI = np.array([0.0, np.pi / 2, np.pi, 2.5])


def lambertian_like(q, q0):
    """BDRF callables reproducing tabulated modes on the quadrature grid (tables come from synthetic.py)."""
    def mk(m):
        def f(mu, neg_mup):
            return q0[m][:, None] if len(np.atleast_1d(neg_mup)) == 1 else q[m]
        return f
    return [mk(m) for m in range(q.shape[0])]


def run(name, cfg, ncol, **extra):
    out = {}
    for i in range(ncol):
        kw = synthetic.column_kwargs(cfg, i)
        kw.update(extra)
        if "bdrf_q" in cfg:
            kw["BDRF_Fourier_modes"] = lambertian_like(cfg["bdrf_q"][i], cfg["bdrf_q0"][i])
        tau_arr = kw["tau_arr"]
        mids = 0.5 * (np.concatenate(([0.0], tau_arr[:-1])) + tau_arr)
        tau_pts = np.sort(np.concatenate(([0.0], tau_arr, mids)))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mu_arr, Fp, Fm, u0, u = PythonicDISORT.pydisort(**kw)
        out[f"c{i}.tau_pts"] = tau_pts
        out[f"c{i}.u"] = u(tau_pts, PHI)
        out[f"c{i}.u0"] = u0(tau_pts)
        out[f"c{i}.flux_up"] = Fp(tau_pts)
        fd = Fm(tau_pts)
        out[f"c{i}.flux_down_diffuse"], out[f"c{i}.flux_down_direct"] = fd
        print(name, i, "done", flush=True)
    out["phi"] = PHI
    out["ncol"] = np.array(ncol)
    os.makedirs(os.path.join(HERE, "synth"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "synth", name + ".npz"), **out)


def run_single(name, kw, tau_pts):
    """One reference call -> inputs are re-created by the test from `literal_cases()` below."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu_arr, Fp, Fm, u0, u = PythonicDISORT.pydisort(**kw)
    out = {"tau_pts": tau_pts, "phi": PHI, "u": u(tau_pts, PHI), "u0": u0(tau_pts), "flux_up": Fp(tau_pts)}
    out["flux_down_diffuse"], out["flux_down_direct"] = Fm(tau_pts)
    os.makedirs(os.path.join(HERE, "synth"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "synth", name + ".npz"), **out)
    print(name, "done", flush=True)


def cloud_c1_cases():
    """BASELINE.json configs[1] as literally worded -- Test Problem 5 (Cloud C.1, one layer of optical depth 64, beam
    source) at 32 streams, delta-M with f = the 32nd moment, Nakajima-Tanaka corrections on -- for both single-scattering
    albedos of the reference's 5a / 5b.  The 300 Cloud C.1 moments are data: they are read from the keyword arguments
    captured in tests/golden/ref/5a.npz (make_reference_goldens.py) and stored in the fixture next to the outputs."""
    sys.path.insert(0, os.path.dirname(HERE))
    import goldens
    leg = np.atleast_2d(goldens.load("5a")[0]["kwargs"]["Leg_coeffs_all"])
    cases = {}
    for tag, omega in (("a", 1 - 1e-6), ("b", 0.9)):
        kw = dict(tau_arr=np.array([64.0]), omega_arr=np.array([omega]), NQuad=32, Leg_coeffs_all=leg, mu0=1.0, I0=np.pi,
                  phi0=np.pi, f_arr=np.array([leg[0, 32]]), NT_cor=True)
        cases["cfg2_q32_cloud_" + tag] = (kw, np.array([0.0, 3.2, 6.4, 12.8, 32.0, 48.0, 64.0]))
    return cases


def run_cloud(name, kw, tau_pts):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu_arr, Fp, Fm, u0, u = PythonicDISORT.pydisort(**{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
    out = {"tau_pts": tau_pts, "phi": PHI, "u": u(tau_pts, PHI), "u0": u0(tau_pts), "flux_up": Fp(tau_pts),
           "Leg_coeffs_all": kw["Leg_coeffs_all"], "omega": kw["omega_arr"]}
    out["flux_down_diffuse"], out["flux_down_direct"] = Fm(tau_pts)
    np.savez_compressed(os.path.join(HERE, "synth", name + ".npz"), **out)
    print(name, "done", flush=True)


if __name__ == "__main__":
    os.makedirs(os.path.join(HERE, "synth"), exist_ok=True)
    if "--cfg5-only" in sys.argv:  # after a change of synthetic.cfg5_columns
        run("cfg5", synthetic.cfg5_columns(8), 8)
        sys.exit(0)
    if "--many-streams-deep-only" in sys.argv:  # round 5: the 66 ... 128-stream workloads that are TIMED, at their full depth
        for name, (maker_kw, nf, ncol) in synthetic.many_stream_deep_cases().items():
            run(name, synthetic.cfg4_columns(ncol, **maker_kw), ncol, NFourier=nf)
        sys.exit(0)
    if "--many-streams-only" in sys.argv:  # 72 / 96 / 128 streams (round 3: the 64-stream cap went)
        for name, (kw, tau_pts) in synthetic.many_stream_cases().items():
            run_single(name, kw, tau_pts)
        sys.exit(0)
    for name, (kw, tau_pts) in synthetic.literal_cases().items():
        run_single(name, kw, tau_pts)
    for name, (kw, tau_pts) in synthetic.many_stream_cases().items():
        run_single(name, kw, tau_pts)
    for name, (kw, tau_pts) in cloud_c1_cases().items():
        run_cloud(name, kw, tau_pts)
    # column counts: cfg4 64, cfg5 8 (twice the survey's 4)
    run("cfg4", synthetic.cfg4_columns(64), 64)
    run("cfg3_big", synthetic.cfg3_columns(4, big=True), 4)
    run("cfg3_small", synthetic.cfg3_columns(4, big=False), 4)
    run("cfg5", synthetic.cfg5_columns(8), 8)
    for name, (maker_kw, nf, ncol) in synthetic.many_stream_deep_cases().items():
        run(name, synthetic.cfg4_columns(ncol, **maker_kw), ncol, NFourier=nf)
