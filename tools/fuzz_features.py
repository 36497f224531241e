#!/usr/bin/env python3
"""Random sweep on the GPU box over the remaining entry points: batched Nakajima-Tanaka corrections against the oracle,
Fourier-mode shards adding up to the unsharded result, layer shards stitched by the boundary-condition solve against the
one-piece solve, device-side BDRF integration against host tables of the same reflectance.
Usage: python tools/fuzz_features.py [ncases]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "pythonic-disort_amd")]
import pydisort_amd
from pydisort_amd import synthetic, subroutines
from pydisort_amd._engine import Plan
from oracle import disort_oracle as O
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
t0 = time.time()


def base(rng, C, L, NQuad, nall):
    g = rng.uniform(0.3, 0.9, (C, L))
    return dict(tau_arr=np.cumsum(10.0 ** rng.uniform(-1.5, 0.7, (C, L)), axis=1), omega_arr=rng.uniform(0.2, 0.99, (C, L)), NQuad=NQuad,
                Leg_coeffs_all=g[:, :, None] ** np.arange(nall)[None, None, :], mu0=rng.uniform(0.15, 1.0, C), I0=rng.uniform(0.5, 3.0, C),
                phi0=rng.uniform(0, 6.0, C), f_arr=g**NQuad), g


for k in range(n):
    rng = np.random.default_rng([55, k])
    NQuad = int(rng.choice([4, 8, 12, 16, 24, 32, 48, 64]))
    C = int(rng.integers(1, 6))
    L = int(rng.integers(1, 6))
    tag = f"case {k}: NQuad {NQuad} C {C} L {L}"
    try:
        # (a) Nakajima-Tanaka corrections, batch vs oracle
        cfg, g = base(rng, C, L, NQuad, NQuad + int(rng.integers(8, 60)))
        M = int(rng.integers(1, min(NQuad, 6) + 1))
        cfg["NFourier"] = M
        _, sol = pydisort_amd.pydisort_batch(NT_cor=True, **cfg)
        tau = np.sort(rng.uniform(0, 1, (C, 3)), axis=1) * cfg["tau_arr"][:, -1:]
        phi = rng.uniform(0, 6.28, 3)
        u = sol.u(tau, phi)
        for i in range(C):
            kw = synthetic.column_kwargs(cfg, i)
            kw.update(NFourier=M, NT_cor=True)
            ur = O.pydisort(**kw)[4](tau[i], phi)
            e = np.max(np.abs(u[i] - ur)) / max(np.max(np.abs(ur)), 1e-300)
            if not e < 2e-8:
                bad += 1; print(tag, f"NT corrections: column {i} vs oracle {e:.2e}", flush=True)
        # (b) mode shards add up
        G = int(rng.integers(1, M + 1))
        _, full = pydisort_amd.pydisort_batch(**cfg)
        uf, u0f, ff = full.u(tau, phi), full.u0(tau), full.flux_up(tau)
        acc, acc0, accf = 0.0, 0.0, 0.0
        for r in range(G):
            _, part = pydisort_amd.pydisort_batch(mode_shard=(r, G), **cfg)
            acc, acc0, accf = acc + part.u(tau, phi), acc0 + part.u0(tau), accf + part.flux_up(tau)
        e = max(np.max(np.abs(acc - uf)), np.max(np.abs(acc0 - u0f)), np.max(np.abs(accf - ff))) / max(np.max(np.abs(uf)), 1e-300)
        if not e < 1e-13:
            bad += 1; print(tag, f"mode shards (G = {G}, M = {M}) do not add up: {e:.2e}", flush=True)
        # (c) layer shards
        if L >= 2:
            plan = full.plan
            want = plan.evaluate(tau, phi)
            Gl = int(rng.integers(2, L + 1))
            cnt = -(-L // Gl)
            Plan.comm_preload()
            plan.comm_init(Plan.comm_unique_id(), 0, 1)
            other, _ = base(rng, C, L, NQuad, NQuad + 1)
            other["NFourier"] = M
            _, osol = pydisort_amd.pydisort_batch(_defer_solve=True, **other)
            plan.set_columns(osol.plan.prep); plan.solve()          # wipe every intermediate
            _, again = pydisort_amd.pydisort_batch(_defer_solve=True, **cfg)
            plan.set_columns(again.plan.prep)
            for first in range(0, L, cnt):
                plan.solve_layers(first, min(cnt, L - first))
            plan.solve_bc()
            got = plan.evaluate(tau, phi)
            e = np.max(np.abs(got["u"] - want["u"])) / max(np.max(np.abs(want["u"])), 1e-300)
            if not e < 1e-11:  # (not bit-equal: a wavefront sweeps until the slowest of ITS problems is done, and the shards regroup them)
                bad += 1; print(tag, f"layer shards ({Gl} pieces) vs one piece: {e:.2e}", flush=True)
        # (d) device-side BDRF integration vs host tables
        if NQuad <= 32:
            a, b, c = rng.uniform(0.05, 0.3), rng.uniform(0.0, 0.5), rng.uniform(0.0, 0.3)
            rho = lambda mu, mup, dphi: a * (1 + b * mu * mup + c * np.sqrt(1 - mu**2) * np.sqrt(1 - mup**2) * np.cos(dphi)) + 0 * dphi
            one = {kk: (v[:1] if isinstance(v, np.ndarray) else v) for kk, v in cfg.items()}
            nb = min(M, 2)
            sq, s0 = subroutines.sample_BDRF(rho, NQuad, float(one["mu0"][0]), nphi=64)
            _, sd = pydisort_amd.pydisort_batch(bdrf_samples=(sq[None], s0[None]), NBDRF=nb, **one)
            mu = sd.mu_arr[: NQuad // 2]
            modes = [lambda m_, n_: a * (1 + b * np.outer(m_, n_)), lambda m_, n_: a * c * np.outer(np.sqrt(1 - m_**2), np.sqrt(1 - np.asarray(n_) ** 2))][:nb]
            kw = synthetic.column_kwargs(one, 0)
            kw.update(NFourier=M, BDRF_Fourier_modes=modes)
            ref = O.pydisort(**kw)
            e = np.max(np.abs(sd.u(tau[:1], phi)[0] - ref[4](tau[0], phi))) / max(np.max(np.abs(ref[4](tau[0], phi))), 1e-300)
            if not e < 1e-8:
                bad += 1; print(tag, f"device BDRF integration vs host modes: {e:.2e}", flush=True)
    except Exception as e:
        bad += 1; print(tag, "EXCEPTION", type(e).__name__, str(e)[:200], flush=True)
print(f"{n} cases, {bad} findings, {time.time() - t0:.0f} s")
