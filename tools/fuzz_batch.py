#!/usr/bin/env python3
"""Random sweep on the GPU box over the BATCH entry points: random stream counts, layer counts, mode counts, source mixes,
window sizes and evaluation points; a batch must equal its columns solved one by one (bit for bit), a windowed plan the
one-window plan (bit for bit), the raw-input streamed path both to rounding, and two columns of every batch the oracle --
also for the antiderivative closures.  Usage: python tools/fuzz_batch.py [nbatches]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd
from pydisort_amd import synthetic
from oracle import disort_oracle as O
warnings.simplefilter("ignore")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
t0 = time.time()
only = int(os.environ["FUZZ_ONLY"]) if os.environ.get("FUZZ_ONLY") else None  # FUZZ_ONLY=1552: that batch alone, with per-column detail
for b in range(nb):
    if only is not None and b != only:
        continue
    rng = np.random.default_rng([77, b])
    NQuad = int(rng.choice([2, 4, 6, 8, 12, 16, 18, 24, 32, 40, 64, 72]))
    N = NQuad // 2
    C = int(rng.integers(1, 40 if NQuad <= 32 else 8))
    L = int(rng.integers(1, 7))
    M = int(rng.integers(1, min(NQuad, 8) + 1))
    g = rng.uniform(0.0, 0.88, (C, L))
    cfg = dict(tau_arr=np.cumsum(10.0 ** rng.uniform(-2.5, 0.8, (C, L)), axis=1), omega_arr=rng.uniform(0.0, 0.99, (C, L)), NQuad=NQuad,
               Leg_coeffs_all=g[:, :, None] ** np.arange(NQuad + 1)[None, None, :], mu0=rng.uniform(0.1, 1.0, C), I0=rng.uniform(0.5, 3.0, C),
               phi0=rng.uniform(0, 6.0, C), NFourier=M)
    if rng.random() < 0.6:
        cfg["f_arr"] = g**NQuad
    if rng.random() < 0.4:
        cfg["s_poly_coeffs"] = rng.uniform(0, 1, (C, L, int(rng.integers(1, 4))))
    if rng.random() < 0.4:
        cfg["b_pos"] = rng.uniform(0, 1, C) if rng.random() < 0.5 else rng.uniform(0, 1, (C, N))
    if rng.random() < 0.4:
        cfg["b_neg"] = float(rng.uniform(0, 1))
    if rng.random() < 0.4 and N > 1:  # (N = 1: the per-column callable below could not tell the quadrature call from the mu0 call)
        nbm = int(rng.integers(1, min(M, 3) + 1))
        cfg["bdrf_q"] = rng.uniform(0.0, 0.3, (C, nbm, N, N)) * 0.5 ** np.arange(nbm)[None, :, None, None]
        cfg["bdrf_q0"] = rng.uniform(0.0, 0.3, (C, nbm, N)) * 0.5 ** np.arange(nbm)[None, :, None]
    if rng.random() < 0.15:
        cfg["I0"] = np.zeros(C)
        if not any(k in cfg for k in ("s_poly_coeffs", "b_pos", "b_neg")):
            cfg["b_neg"] = 0.3
    ntau = int(rng.integers(1, 6))
    tauL = cfg["tau_arr"][:, -1]
    tau = np.sort(rng.uniform(0, 1, (C, ntau)), axis=1) * tauL[:, None]
    tau[:, 0] = np.where(rng.random(C) < 0.3, 0.0, tau[:, 0])
    tau[:, -1] = np.where(rng.random(C) < 0.3, tauL, tau[:, -1])
    if L > 1 and ntau > 2:
        tau[:, 1] = cfg["tau_arr"][:, 0]  # exactly on an interface
        tau = np.sort(tau, axis=1)
    phi = rng.uniform(0, 6.28, int(rng.integers(1, 4)))
    tag = f"batch {b}: NQuad {NQuad} C {C} L {L} M {M} keys {sorted(set(cfg) - {'tau_arr', 'omega_arr', 'NQuad', 'Leg_coeffs_all', 'mu0', 'I0', 'phi0', 'NFourier'})}"
    try:
        _, sol = pydisort_amd.pydisort_batch(**cfg)
        u, fu, fd = sol.u(tau, phi), sol.flux_up(tau), sol.flux_down(tau)
        ua = sol.u(tau, phi, True)
        # windowed plan
        wc = int(rng.integers(1, C + 1))
        _, solw = pydisort_amd.pydisort_batch(work_columns=wc, **cfg)
        if not (np.array_equal(solw.u(tau, phi), u) and np.array_equal(solw.flux_up(tau), fu)):
            bad += 1; print(tag, "WINDOWED plan differs (work_columns %d)" % wc, flush=True)
        # columns one by one
        for i in rng.choice(C, min(C, 3), replace=False):
            kw = synthetic.column_kwargs(cfg, i)
            kw["NFourier"] = M
            if "bdrf_q" in cfg:
                q, q0 = cfg["bdrf_q"][i], cfg["bdrf_q0"][i]
                kw["BDRF_Fourier_modes"] = [(lambda mu, nmup, m=m: q0[m][:, None] if len(np.atleast_1d(nmup)) == 1 else q[m]) for m in range(q.shape[0])]
            one = pydisort_amd.pydisort(**kw)
            shp = u[i].shape  # (the drop-in closures squeeze singleton axes, as the reference's do)
            ou = np.reshape(one[4](tau[i], phi), shp)
            if not (np.array_equal(ou, u[i]) and np.array_equal(np.reshape(one[1](tau[i]), fu[i].shape), fu[i])):
                d = np.max(np.abs(ou - u[i])) / max(np.max(np.abs(u[i])), 1e-300)
                if d > 1e-12:
                    bad += 1; print(tag, "column %d alone differs from the batch: %.2e" % (i, d), flush=True)
            ref = O.pydisort(**kw)
            ur = np.reshape(ref[4](tau[i], phi), shp)
            ura = np.reshape(ref[4](tau[i], phi, True), shp)
            sc = max(np.max(np.abs(ur)), np.max(np.abs(ref[3](tau[i]))), 1e-300)
            e1 = np.max(np.abs(u[i] - ur)) / sc
            e2 = np.max(np.abs(ua[i] - ura)) / max(np.max(np.abs(ura)), sc)
            e3 = np.max(np.abs(fu[i] - np.reshape(ref[1](tau[i]), fu[i].shape))) / sc
            e4 = np.max(np.abs(fd[0][i] - np.reshape(ref[2](tau[i])[0], fu[i].shape))) / sc
            tol = 1e-7 if NQuad > 64 else 2e-8
            if not (e1 < tol and e2 < tol and e3 < tol and e4 < tol):
                bad += 1; print(tag, "column %d vs oracle: u %.2e antiderivative %.2e flux_up %.2e flux_down %.2e" % (i, e1, e2, e3, e4), flush=True)
        if True:
            res = pydisort_amd.solve_columns_streamed({k: v for k, v in cfg.items()}, tau, phi, chunk_columns=max(1, wc))
            d = np.max(np.abs(res["u"] - u)) / max(np.max(np.abs(u)), 1e-300)
            if d > 1e-11:
                bad += 1; print(tag, "streamed raw path differs: %.2e" % d, flush=True)
            if only is not None:  # per column: both preparations against the oracle
                for i in range(C):
                    kw = synthetic.column_kwargs(cfg, i)
                    kw["NFourier"] = M
                    ur = np.reshape(O.pydisort(**kw)[4](tau[i], phi), u[i].shape)
                    sc = max(np.max(np.abs(ur)), 1e-300)
                    print("  column %d: numpy-prepared vs oracle %.2e, raw vs oracle %.2e, raw vs numpy-prepared %.2e, mu0 %.6f, max omega %.6f"
                          % (i, np.max(np.abs(u[i] - ur)) / sc, np.max(np.abs(res["u"][i] - ur)) / sc, np.max(np.abs(res["u"][i] - u[i])) / sc,
                             cfg["mu0"][i], cfg["omega_arr"][i].max()), flush=True)
        sol.plan.close(); solw.plan.close()
    except Exception as e:
        bad += 1; print(tag, "EXCEPTION", type(e).__name__, str(e)[:160], flush=True)
print(f"{nb} batches, {bad} findings, {time.time() - t0:.0f} s")
