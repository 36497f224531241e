#!/usr/bin/env python3
"""High-precision (mpmath, 40 digits) solution of the discrete-ordinate problem (every Fourier mode) for a
beam + Dirichlet atmosphere, straight from the ODE system of SURVEY Appendix A.2-A.4 (full 2N x 2N eigenproblem per
layer, dense boundary-condition system) -- no symmetrisation, no scaling tricks.  Used to decide who is right when the
float64 oracle (the reference's algorithm) and the HIP path disagree at the 1e-7 level on near-conservative layers
(omega = 1 - 1e-6): writes u0 at the layer interfaces to an .npz that tools/hp_compare.py reads on the GPU box."""
import os, sys
import numpy as np
import mpmath as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
from oracle import disort_oracle as O

mp.mp.dps = 40


from hp_cases import harsh_case, benign_case, intensity_case, PHI  # noqa: E402


def solve_hp(kw, m=0):
    """-> u^m at the layer interfaces [Q, L+1] (times the rescale factor), Fourier mode m."""
    p = O.prepare(**kw)
    L, N, P = p["L"], p["N"], p["P"]
    mu = [mp.mpf(float(x)) for x in p["mu"]]
    w = [mp.mpf(float(x)) for x in p["W"]]
    # associated Legendre functions P_l^m at +mu_i and at -mu0, with the weight sqrt((l-m)!/(l+m)!) folded in (the
    # reference's poch factor (:64-65) split evenly over the two factors of every product; signs cancel in pairs)
    def leg(x, n):
        return [mp.mpf(0) if l < m else mp.legenp(l, m, x, type=2) * mp.sqrt(mp.factorial(l - m) / mp.factorial(l + m))
                for l in range(n)]
    Y = [leg(m_, P) for m_ in mu]            # Y[i][l]
    mu0 = mp.mpf(p["mu0"])
    Y0 = leg(-mu0, P)
    ts = [mp.mpf(float(x)) for x in p["tau_s0"]]
    Q = 2 * N
    Gs, Ks, Bs = [], [], []
    for l in range(L):
        om = mp.mpf(float(p["omega_s"][l]))
        wl = [mp.mpf(float(x)) for x in p["wleg"][l]]
        Dp = mp.zeros(N)
        Dm = mp.zeros(N)
        for i in range(N):
            for j in range(N):
                sp = sm = mp.mpf(0)
                for ell in range(P):
                    t = om / 2 * wl[ell] * Y[i][ell] * Y[j][ell]
                    sp += t
                    sm += t * (-1) ** (ell - m)
                Dp[i, j], Dm[i, j] = sp, sm
        A = mp.zeros(Q)
        for i in range(N):
            for j in range(N):
                al = (Dp[i, j] * w[j] - (1 if i == j else 0)) / mu[i]
                be = Dm[i, j] * w[j] / mu[i]
                A[i, j], A[i, N + j], A[N + i, j], A[N + i, N + j] = -al, -be, be, al
        ev, G = mp.eig(A)
        ev = [mp.re(e) for e in ev]
        G = G.apply(mp.re)
        # beam particular solution: (A + I/mu0) B = Xtilde   
        X = mp.zeros(Q, 1)
        for i in range(N):
            xp = xm = mp.mpf(0)
            for ell in range(P):
                t = mp.mpf(float(p["I0_4pi"])) * (1 if m == 0 else 2) * om * wl[ell] * Y0[ell] * Y[i][ell]
                xp += t
                xm += t * (-1) ** (ell - m)
            X[i], X[N + i] = xp / mu[i], -xm / mu[i]
        Bv = mp.lu_solve(A + mp.eye(Q) / mu0, X)
        Gs.append(G); Ks.append(ev); Bs.append(Bv)
    # boundary-condition system for the coefficients: u_l(t) = G_l diag(exp(K_l (t - ref_l))) C_l + B_l exp(-t/mu0)
    # (each exponential referenced to the layer boundary where it is <= 1, as the reference does)
    def mode(l, j, t):
        k = Ks[l][j]
        ref = ts[l + 1] if k > 0 else ts[l]
        return mp.e ** (k * (t - ref))
    n = Q * L
    Amat = mp.zeros(n)
    rhs = mp.zeros(n, 1)
    row = 0
    bneg = mp.mpf(float(p["b_neg"][0, m])); bpos = mp.mpf(float(p["b_pos"][0, m]))
    for i in range(N):  # top: downward streams
        for j in range(Q):
            Amat[row, j] = Gs[0][N + i, j] * mode(0, j, ts[0])
        rhs[row] = bneg - Bs[0][N + i] * mp.e ** (-ts[0] / mu0)
        row += 1
    for l in range(L - 1):
        t = ts[l + 1]
        for i in range(Q):
            for j in range(Q):
                Amat[row, l * Q + j] = Gs[l][i, j] * mode(l, j, t)
                Amat[row, (l + 1) * Q + j] = -Gs[l + 1][i, j] * mode(l + 1, j, t)
            rhs[row] = (Bs[l + 1][i] - Bs[l][i]) * mp.e ** (-t / mu0)
            row += 1
    for i in range(N):  # bottom: upward streams
        for j in range(Q):
            Amat[row, (L - 1) * Q + j] = Gs[L - 1][i, j] * mode(L - 1, j, ts[L])
        rhs[row] = bpos - Bs[L - 1][i] * mp.e ** (-ts[L] / mu0)
        row += 1
    Cc = mp.lu_solve(Amat, rhs)
    # u0 at the interfaces (top of layer 0, then the bottom of every layer)
    out = np.zeros((Q, L + 1))
    for ti in range(L + 1):
        l = 0 if ti == 0 else ti - 1
        t = ts[ti]
        for i in range(Q):
            v = Bs[l][i] * mp.e ** (-t / mu0)
            for j in range(Q):
                v += Gs[l][i, j] * mode(l, j, t) * Cc[l * Q + j]
            out[i, ti] = float(v * mp.mpf(float(p["rescale"])))
    return out


if __name__ == "__main__":
    res = {}
    for name, kw in (("benign", benign_case()), ("harsh", harsh_case())):
        hp = solve_hp(kw)
        mu_arr, fu, fd, u0 = O.pydisort(**kw)
        tau = np.concatenate(([0.0], kw["tau_arr"]))
        orc = u0(tau)
        scale = np.max(np.abs(hp))
        print(name, "oracle vs high precision: max rel err %.2e" % (np.max(np.abs(orc - hp)) / scale), flush=True)
        res[name] = hp
        res[name + "_oracle"] = orc
    # full intensity: u(tau, phi) = sum_m u^m cos(m (phi0 - phi))
    kw = intensity_case()
    M = kw["NQuad"]
    um = [solve_hp(kw, m) for m in range(M)]
    u = sum(um[m][:, :, None] * np.cos(m * (kw["phi0"] - PHI))[None, None, :] for m in range(M))
    mu_arr, fu, fd, u0, uf = O.pydisort(**kw)
    tau = np.concatenate(([0.0], kw["tau_arr"]))
    orc = uf(tau, PHI)
    print("intensity oracle vs high precision: max rel err %.2e" % (np.max(np.abs(orc - u)) / np.max(np.abs(u))), flush=True)
    res["intensity"] = u
    res["intensity_oracle"] = orc
    np.savez(os.path.join(ROOT, "tests", "golden", "hp_truth_m0.npz"), **res)
