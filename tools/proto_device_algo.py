#!/usr/bin/env python3
"""Development prototype (NumPy) of the algorithm the HIP kernels implement -- NOT product code.

It mirrors, stage by stage, what pythonic-disort_amd/csrc/*.hip does on the device
(normalised Legendre recurrences, symmetrised eigenproblem solved by parallel-order
Jacobi, spectral beam particular solution, block-bidiagonal pivoted LU for the
boundary-condition system) so that the math can be checked against the oracle on
the CPU before a kernel is written.  Run:  python tools/proto_device_algo.py
"""
import os
import sys
from math import factorial, pi

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import disort_oracle as O  # noqa: E402


def ybar(m, P, x):
    """sqrt((l-m)!/(l+m)!) |P_l^m(x)|-type normalised functions, l = m..P-1 (sign (-1)^m dropped)."""
    x = np.asarray(x, float)
    Y = np.zeros((P - m, len(x)))
    v = np.ones_like(x)
    for j in range(1, m + 1):
        v = v * np.sqrt((2 * j - 1) / (2 * j)) * np.sqrt(1 - x * x)
    Y[0] = v
    if P - m > 1:
        Y[1] = np.sqrt(2 * m + 1) * x * v
    for l in range(m + 1, P - 1):
        Y[l + 1 - m] = ((2 * l + 1) * x * Y[l - m] - np.sqrt((l + m) * (l - m)) * Y[l - 1 - m]) \
            / np.sqrt((l + 1 - m) * (l + 1 + m))
    return Y


def jacobi_xor(H, sweeps=12, tol=1e-30):
    """Parallel-order cyclic Jacobi; N must be a power of two; pairs (j, j^t), t = 1..N-1."""
    H = H.copy()
    n = H.shape[0]
    Z = np.eye(n)
    used = 0
    for s in range(sweeps):
        off = np.sum((H - np.diag(np.diag(H))) ** 2)
        if off <= tol * np.sum(np.diag(H) ** 2):
            break
        used += 1
        for t in range(1, n):
            J = np.eye(n)
            for p in range(n):
                q = p ^ t
                if p > q:
                    continue
                if H[p, q] == 0.0:
                    continue
                zeta = (H[q, q] - H[p, p]) / (2 * H[p, q])
                tt = np.sign(zeta) / (abs(zeta) + np.sqrt(1 + zeta * zeta)) if zeta != 0 else 1.0
                c = 1 / np.sqrt(1 + tt * tt)
                sn = tt * c
                J[p, p] = c
                J[q, q] = c
                J[p, q] = sn
                J[q, p] = -sn
            H = J.T @ H @ J
            Z = Z @ J
    return np.diag(H).copy(), Z, used


def eig_stage(p, use_jacobi=True, NP=None):
    """-> Gp, Gm [M,L,N,N], k [M,L,N], Bp, Bm [M,L,N], zneg [L,N] (m=0), dq [L,Ns,2N]."""
    L, N, P, M = p["L"], p["N"], p["P"], p["M"]
    mu, W, mu0 = p["mu"], p["W"], p["mu0"]
    Gp = np.zeros((M, L, N, N))
    Gm = np.zeros((M, L, N, N))
    kk = np.zeros((M, L, N))
    Bp = np.zeros((M, L, N))
    Bm = np.zeros((M, L, N))
    zneg = np.zeros((L, N))
    S = np.sqrt(W / mu)
    T = np.sqrt(W * mu)
    sweeps = []
    for m in range(M):
        Y = ybar(m, P, mu)
        Y0 = ybar(m, P, np.array([-mu0]))[:, 0] if p["beam"] else None
        par = (np.arange(P - m) % 2)
        for l in range(L):
            c = 0.5 * p["omega_s"][l] * p["wleg"][l, m:]
            if not np.any(np.abs(c) > 1e-8):
                Gm[m, l] = np.eye(N)
                kk[m, l] = 1 / mu
                if m == 0:
                    zneg[l] = -1 / mu
                continue
            Ae = 2 * (Y[par == 0].T * c[par == 0]) @ Y[par == 0]
            Ao = 2 * (Y[par == 1].T * c[par == 1]) @ Y[par == 1]
            Pm = np.diag(1 / mu) - S[:, None] * Ae * S[None, :]
            Qm = np.diag(1 / mu) - S[:, None] * Ao * S[None, :]
            Lc = np.linalg.cholesky(Pm)
            H = Lc.T @ Qm @ Lc
            if use_jacobi:
                n2 = 1 << (N - 1).bit_length()
                Hp = np.eye(n2) * 0.25
                Hp[:N, :N] = H
                k2, Zp, used = jacobi_xor(Hp)
                sweeps.append(used)
                k2, Z = k2[:N], Zp[:N, :N]
            else:
                k2, Z = np.linalg.eigh(H)
            k = np.sqrt(k2)
            Vt = np.linalg.solve(Lc.T, Z) / T[:, None]
            Ut = -(Lc @ Z) / k[None, :] / T[:, None]
            Gp[m, l], Gm[m, l], kk[m, l] = Vt + Ut, Vt - Ut, k
            if m == 0:
                zneg[l] = -0.5 * k * (Z.T @ np.linalg.solve(Lc, T / mu))
            if p["beam"]:
                xi = p["I0_4pi"] * (2 - (m == 0)) * p["omega_s"][l] * p["wleg"][l, m:] * Y0
                Xe = xi[par == 0] @ Y[par == 0]
                Xo = xi[par == 1] @ Y[par == 1]
                xs, xd = 2 * Xo / mu, 2 * Xe / mu
                rhat = T * xs / mu0 - Qm @ (T * xd)
                shat = np.linalg.solve(Lc.T, Z @ ((Z.T @ (Lc.T @ rhat)) / (1 / mu0**2 - k2)))
                dhat = mu0 * (T * xd - Pm @ shat)
                s, d = shat / T, dhat / T
                Bp[m, l], Bm[m, l] = 0.5 * (s + d), 0.5 * (s - d)
    dq = np.zeros((L, max(p["Ns"], 1), 2 * N))
    if p["iso"]:
        for l in range(L):
            k = kk[0, l]
            for q in range(p["Ns"]):
                bneg = sum(factorial(j) / factorial(q) * p["s_s"][l, j] / (-k) ** (j - q + 1) for j in range(q, p["Ns"]))
                bpos = sum(factorial(j) / factorial(q) * p["s_s"][l, j] / k ** (j - q + 1) for j in range(q, p["Ns"]))
                a, b = zneg[l] * bneg, -zneg[l] * bpos
                dq[l, q, :N] = Gp[0, l] @ a + Gm[0, l] @ b
                dq[l, q, N:] = Gm[0, l] @ a + Gp[0, l] @ b
    return Gp, Gm, kk, Bp, Bm, zneg, dq, sweeps


def bc_stage(p, Gp, Gm, kk, Bp, Bm, dq):
    """Block forward elimination with partial pivoting inside each [3N x 2N] panel, then back substitution."""
    L, N, M = p["L"], p["N"], p["M"]
    Q = 2 * N
    mu, W, mu0, ts0 = p["mu"], p["W"], p["mu0"], p["tau_s0"]
    Cc = np.zeros((M, L, Q))
    vpoly = lambda l, t: sum(dq[l, q] * t**q for q in range(p["Ns"])) if p["iso"] else np.zeros(Q)
    for m in range(M):
        E = np.exp(-kk[m] * np.diff(ts0)[:, None])
        Bfull = np.concatenate((Bp[m], Bm[m]), axis=1)

        def Pblk(l):  # [G_l[:, :N] E_l, G_l[:, N:]]
            return np.block([[Gp[m, l] * E[l], Gm[m, l]], [Gm[m, l] * E[l], Gp[m, l]]])

        def Qblk(l):  # [G_l[:, :N], G_l[:, N:] E_l]
            return np.block([[Gp[m, l], Gm[m, l] * E[l]], [Gm[m, l], Gp[m, l] * E[l]]])

        # carry = top BC rows
        carry = np.concatenate((Gm[m, 0], Gp[m, 0] * E[0]), axis=1)
        cr = p["b_neg"][:, m] - Bfull[0, N:] - (vpoly(0, 0.0)[N:] if m == 0 else 0)
        Us, Fs, ys = [], [], []
        for l in range(L - 1):
            tb = ts0[l + 1]
            rhs_i = (Bfull[l + 1] - Bfull[l]) * np.exp(-tb / mu0)
            if m == 0:
                rhs_i = rhs_i + vpoly(l + 1, tb) - vpoly(l, tb)
            panel = np.zeros((3 * N, 2 * Q + 1))
            panel[:N, :Q] = carry
            panel[:N, -1] = cr
            panel[N:, :Q] = Pblk(l)
            panel[N:, Q:2 * Q] = -Qblk(l + 1)
            panel[N:, -1] = rhs_i
            active = np.ones(3 * N, bool)
            order = []
            for kcol in range(Q):
                cand = np.where(active, np.abs(panel[:, kcol]), -1.0)
                pr = int(np.argmax(cand))
                active[pr] = False
                order.append(pr)
                f = panel[:, kcol] / panel[pr, kcol]
                f[~active] = 0.0
                panel -= f[:, None] * panel[pr][None, :]
            Us.append(panel[order, :Q])
            Fs.append(panel[order, Q:2 * Q])
            ys.append(panel[order, -1])
            rest = np.where(active)[0]
            carry = panel[rest, Q:2 * Q]
            cr = panel[rest, -1]
        # last layer: carry + bottom BC
        l = L - 1
        att = np.exp(-ts0[-1] / mu0)
        vb = vpoly(l, ts0[-1]) if m == 0 else np.zeros(Q)
        if len(p["bdrf"]) > m:
            R = (1 + (m == 0)) * p["bdrf"][m][0] * (mu * W)[None, :]
            bot = np.concatenate(((Gp[m, l] - R @ Gm[m, l]) * E[l], Gm[m, l] - R @ Gp[m, l]), axis=1)
            br = p["b_pos"][:, m] + (mu0 * p["I0_4pi"] * 4 * p["bdrf"][m][1] + R @ Bfull[l, N:] - Bfull[l, :N]) * att \
                + R @ vb[N:] - vb[:N]
        else:
            bot = np.concatenate((Gp[m, l] * E[l], Gm[m, l]), axis=1)
            br = p["b_pos"][:, m] - Bfull[l, :N] * att - vb[:N]
        x = np.linalg.solve(np.concatenate((carry, bot)), np.concatenate((cr, br)))
        Cc[m, l] = x
        for l in range(L - 2, -1, -1):
            x = np.linalg.solve(np.triu(Us[l]), ys[l] - Fs[l] @ x)
            Cc[m, l] = x
    return Cc


def evaluate(p, Gp, Gm, kk, Bp, Bm, dq, Cc, tau, phi):
    N, M = p["N"], p["M"]
    tau = np.atleast_1d(tau)
    l = np.argmax(tau[:, None] <= p["tau"][None, :], axis=1)
    ts = p["tau_s0"][1:][l] - (p["tau"][l] - tau) * p["scale_tau"][l]
    um = np.zeros((M, 2 * N, len(tau)))
    for t in range(len(tau)):
        ll = l[t]
        for m in range(M):
            en = np.exp(-kk[m, ll] * (ts[t] - p["tau_s0"][ll])) * Cc[m, ll, :N]
            ep = np.exp(-kk[m, ll] * (p["tau_s0"][ll + 1] - ts[t])) * Cc[m, ll, N:]
            um[m, :N, t] = Gp[m, ll] @ en + Gm[m, ll] @ ep + Bp[m, ll] * np.exp(-ts[t] / p["mu0"] if p["beam"] else 0)
            um[m, N:, t] = Gm[m, ll] @ en + Gp[m, ll] @ ep + Bm[m, ll] * np.exp(-ts[t] / p["mu0"] if p["beam"] else 0)
        if p["iso"]:
            um[0, :, t] += sum(dq[ll, q] * ts[t] ** q for q in range(p["Ns"]))
    cosm = np.cos(np.arange(M)[:, None] * (p["phi0"] - np.atleast_1d(phi))[None, :])
    return p["rescale"] * np.einsum("mit,mp->itp", um, cosm)


def main():
    import goldens

    for tid in ["1a", "2a", "6c", "7b", "8b", "9c", "8ARTS_B"]:
        call = goldens.load(tid)[0]
        kw = call["kwargs"]
        p = O.prepare(**kw)
        sol = O.Solution(p)
        st = eig_stage(p)
        Cc = bc_stage(p, *st[:5], st[6])
        tau = np.concatenate(([0.0], p["tau"], 0.37 * p["tau"][:1]))
        phi = np.array([0.0, 1.0, pi])
        got = evaluate(p, *st[:5], st[6], Cc, tau, phi)
        want = sol.u(tau, phi) if not p["only_flux"] else None
        if want is None:
            got = got[:, :, 0]
            want = sol.u0(tau)
        err = np.max(np.abs(got - np.reshape(want, got.shape))) / np.max(np.abs(want))
        print(f"{tid:8s} N={p['N']:2d} L={p['L']:2d} M={p['M']:2d} err={err:.2e} jacobi sweeps max={max(st[7], default=0)}")


if __name__ == "__main__":
    main()


# ------------------------------------------------------------------------------------------------
# Structured block elimination of the BC system (candidate replacement of bc_stage):
# multiply the continuity rows of interface l by G_l^-1 (known in closed form from the eigen stage)
# so that the x_l block becomes diag(E_l, 1); eliminate C+_l with those rows and C-_l with the carry.
# ------------------------------------------------------------------------------------------------
def bc_stage_structured(p, Gp, Gm, kk, Bp, Bm, dq):
    L, N, M = p["L"], p["N"], p["M"]
    Q = 2 * N
    mu, W, mu0, ts0 = p["mu"], p["W"], p["mu0"], p["tau_s0"]
    Cc = np.zeros((M, L, Q))
    vpoly = lambda l, t: sum(dq[l, q] * t**q for q in range(p["Ns"])) if p["iso"] else np.zeros(Q)
    for m in range(M):
        E = np.exp(-kk[m] * np.diff(ts0)[:, None])
        Bfull = np.concatenate((Bp[m], Bm[m]), axis=1)
        Vi = [np.linalg.inv(0.5 * (Gp[m, l] + Gm[m, l])) for l in range(L)]  # device: Z^T L^T T (closed form)
        Ui = [np.linalg.inv(0.5 * (Gp[m, l] - Gm[m, l])) for l in range(L)]
        Ta, Tb = Gm[m, 0].copy(), Gp[m, 0] * E[0]
        t = p["b_neg"][:, m] - Bfull[0, N:] - (vpoly(0, 0.0)[N:] if m == 0 else 0)
        keep = []
        for l in range(L - 1):
            tb = ts0[l + 1]
            r = (Bfull[l + 1] - Bfull[l]) * (np.exp(-tb / mu0) if p["beam"] else 0.0)
            if m == 0:
                r = r + vpoly(l + 1, tb) - vpoly(l, tb)
            # G_l^-1 = 1/4 [[Vi+Ui, Vi-Ui],[Vi-Ui, Vi+Ui]] with V = (Gp+Gm)/2, U = (Gp-Gm)/2
            rs, rd = r[:N] + r[N:], r[:N] - r[N:]
            rho_t = 0.25 * (Vi[l] @ rs + Ui[l] @ rd)
            rho_b = 0.25 * (Vi[l] @ rs - Ui[l] @ rd)
            VV = Vi[l] @ (0.5 * (Gp[m, l + 1] + Gm[m, l + 1]))
            UU = Ui[l] @ (0.5 * (Gp[m, l + 1] - Gm[m, l + 1]))
            Wp, Wq = 0.5 * (VV + UU), 0.5 * (VV - UU)
            sol = np.linalg.solve(Ta, np.concatenate((Tb, t[:, None]), axis=1))
            S, s = sol[:, :N], sol[:, N]
            keep.append((S, s, Wp, Wq, rho_b))
            ES = E[l][:, None] * S
            Ta_n = -(ES @ Wq + Wp)
            Tb_n = -(ES @ Wp + Wq) * E[l + 1][None, :]
            t = rho_t - E[l] * (s - S @ rho_b)
            Ta, Tb = Ta_n, Tb_n
        l = L - 1
        att = np.exp(-ts0[-1] / mu0) if p["beam"] else 0.0
        vb = vpoly(l, ts0[-1]) if m == 0 else np.zeros(Q)
        if len(p["bdrf"]) > m:
            R = (1 + (m == 0)) * p["bdrf"][m][0] * (mu * W)[None, :]
            Ba, Bb = (Gp[m, l] - R @ Gm[m, l]) * E[l], Gm[m, l] - R @ Gp[m, l]
            br = p["b_pos"][:, m] + (mu0 * p["I0_4pi"] * 4 * p["bdrf"][m][1] + R @ Bfull[l, N:] - Bfull[l, :N]) * att \
                + R @ vb[N:] - vb[:N]
        else:
            Ba, Bb = Gp[m, l] * E[l], Gm[m, l]
            br = p["b_pos"][:, m] - Bfull[l, :N] * att - vb[:N]
        sol = np.linalg.solve(Ta, np.concatenate((Tb, t[:, None]), axis=1))
        S, s = sol[:, :N], sol[:, N]
        cp = np.linalg.solve(Bb - Ba @ S, br - Ba @ s)
        cm_ = s - S @ cp
        Cc[m, l] = np.concatenate((cm_, cp))
        for l in range(L - 2, -1, -1):
            S, s, Wp, Wq, rho_b = keep[l]
            nxt_m, nxt_p = Cc[m, l + 1, :N], Cc[m, l + 1, N:] * E[l + 1]
            cp = Wq @ nxt_m + Wp @ nxt_p + rho_b
            Cc[m, l] = np.concatenate((s - S @ cp, cp))
    return Cc


def check_structured(ids=None):
    import goldens
    import warnings
    warnings.simplefilter("ignore")
    ids = ids or goldens.list_ids()
    for tid in ids:
        worst = 0.0
        for call in goldens.load(tid)[:3]:
            p = O.prepare(**call["kwargs"])
            st = eig_stage(p, use_jacobi=False)
            C1 = bc_stage(p, *st[:5], st[6])
            C2 = bc_stage_structured(p, *st[:5], st[6])
            tau = np.concatenate(([0.0], p["tau"], 0.37 * p["tau"][:1]))
            phi = np.array([0.0, 1.0])
            u1 = evaluate(p, *st[:5], st[6], C1, tau, phi)
            u2 = evaluate(p, *st[:5], st[6], C2, tau, phi)
            sc = np.max(np.abs(u1))
            if sc > 0:
                worst = max(worst, np.max(np.abs(u1 - u2)) / sc)
        print(f"{tid:12s} structured vs pivoted-GJ: {worst:.2e}", flush=True)
