#!/usr/bin/env python3
"""NumPy replay of the eigen kernel's one-sided Jacobi iteration on cfg4 problems (CPU; no GPU needed) -- the numbers
behind DESIGN.md section 7(c):

  * sweeps per problem and per Fourier mode (stop rule of the kernel: a sweep is the last one when every pair it met had
    cos^2 <= 1e-14 before its rotation);
  * what the wavefront grouping costs: a wavefront holds four layers of one (column, mode) and sweeps until the slowest is
    done -- mean sweeps per wavefront under the kernel's layer order (ascending omega* / (1 - g*)), under other orders,
    and the floor (problems regrouped freely inside a mode);
  * the convergence law: max cos^2 of sweep k + 1 against that of sweep k, and what a stop at a looser threshold would leave
    behind.

F = L^T R is formed exactly as the kernel forms it (Pm = M^-1 - S 2D_e S = L L^T, Qm = M^-1 - S 2D_o S = R R^T;
_solve_for_gen_and_part_sols.py:123-135 symmetrised).  The replay uses a round-robin pair order, not the kernel's butterfly:
sweep counts differ by a few per cent, the laws do not.

Usage: python3 tools/jacobi_convergence.py [columns]      (default 8; about 10 s)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pythonic-disort_amd"), os.path.join(ROOT, "tools")]
from oracle import disort_oracle as O  # noqa: E402  (input preparation only)
from pydisort_amd import synthetic  # noqa: E402
from proto_device_algo import ybar  # noqa: E402


def F_of_column(kw):
    p = O.prepare(**kw)
    L, N, P, M = p["L"], p["N"], p["P"], p["M"]
    mu, W = p["mu"], p["W"]
    S = np.sqrt(W / mu)
    Fs = np.zeros((M, L, N, N))
    for m in range(M):
        Y = ybar(m, P, mu)
        par = np.arange(P - m) % 2
        for l in range(L):
            c = 0.5 * p["omega_s"][l] * p["wleg"][l, m:]
            Ae = 2 * (Y[par == 0].T * c[par == 0]) @ Y[par == 0]
            Ao = 2 * (Y[par == 1].T * c[par == 1]) @ Y[par == 1]
            Pm = np.diag(1 / mu) - S[:, None] * Ae * S[None, :]
            Qm = np.diag(1 / mu) - S[:, None] * Ao * S[None, :]
            Fs[m, l] = np.linalg.cholesky(Pm).T @ np.linalg.cholesky(Qm)
    return Fs, p


def sweep_maxima(W, sweeps=10):
    """max over the pairs of cos^2 BEFORE the rotation, per sweep and problem: [sweeps, n]."""
    n, N, _ = W.shape
    W = W.copy()
    out = []
    for _ in range(sweeps):
        tau = np.zeros(n)
        order = list(range(N))
        for _ in range(N - 1):
            a, b = np.array(order[:N // 2]), np.array(order[N // 2:][::-1])
            x, y = W[:, :, a], W[:, :, b]
            gam, ax, ay = np.sum(x * y, 1), np.sum(x * x, 1), np.sum(y * y, 1)
            tau = np.maximum(tau, np.max(gam * gam / (ax * ay), 1))
            delta, g2 = ay - ax, 2 * gam
            t = g2 / (delta + np.copysign(np.sqrt(delta * delta + g2 * g2 + 1e-280), delta))
            c = 1 / np.sqrt(1 + t * t)
            sn = t * c
            W[:, :, a], W[:, :, b] = c[:, None, :] * x - sn[:, None, :] * y, sn[:, None, :] * x + c[:, None, :] * y
            order = [order[0]] + [order[-1]] + order[1:-1]
        out.append(tau)
    return np.array(out)


if __name__ == "__main__":
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cfg = synthetic.cfg4_columns(C)
    Fs, om, g = [], [], []
    for i in range(C):
        F, p = F_of_column(synthetic.column_kwargs(cfg, i))
        Fs.append(F.reshape(-1, 16, 16))
        om.append(p["omega_s"])
        g.append(p["wleg"][:, 1] / 3)
    M, L = 32, 20
    T = sweep_maxima(np.concatenate(Fs))
    n = T.shape[1]
    last = np.argmax(T <= 1e-14, axis=0)          # the first sweep whose pairs were all below the tolerance: the last one
    sw = (last + 1).reshape(C, M, L)
    om, g = np.array(om), np.array(g)
    print(f"{n} problems: {sw.mean():.2f} sweeps per problem (max {sw.max()}); by Fourier mode:", np.round(sw.mean((0, 2)), 1))

    def per_wave(order_of):
        tot = 0
        for c in range(C):
            for m in range(M):
                tot += sw[c, m, order_of(c, m)].reshape(5, 4).max(1).sum()
        return tot / (C * M * 5)
    print("sweeps per wavefront (4 layers of one (column, mode), the slowest counts):")
    print(f"  layers in their natural order        {per_wave(lambda c, m: np.arange(L)):.2f}")
    print(f"  kernel's order, omega* / (1 - g*)    {per_wave(lambda c, m: np.argsort(om[c] / np.maximum(1 - g[c], 1e-6))):.2f}")
    print(f"  omega* g*^m                          {per_wave(lambda c, m: np.argsort(om[c] * g[c] ** m)):.2f}")
    print(f"  sorted by the true sweep count       {per_wave(lambda c, m: np.argsort(sw[c, m])):.2f}")
    tot = sum(np.sort(sw[:, m, :].ravel()).reshape(-1, 4).max(1).sum() for m in range(M))
    print(f"  regrouped freely inside a mode       {tot / (C * M * 5):.2f}   (needs per-lane table and mu0 loads)")
    r = np.concatenate([T[k + 1][(T[k] < 1e-3) & (T[k] > 1e-13)] / T[k][(T[k] < 1e-3) & (T[k] > 1e-13)] ** 2 for k in range(T.shape[0] - 1)])
    print("cos^2 of the next sweep / (cos^2)^2: median %.1f, 99.9 %% %.0f, max %.0f" % (np.median(r), np.quantile(r, 0.999), r.max()))
    for thr in (1e-8, 1e-9, 1e-10, 1e-14):
        first = np.argmax(T <= thr, axis=0)
        resid = T[np.minimum(first + 1, T.shape[0] - 1), np.arange(n)]
        print(f"stop after the first sweep with cos^2 <= {thr:g}: {(first + 1).mean():.2f} sweeps, leaves cos <= {np.sqrt(resid.max()):.1e} "
              "(eigenvector error of that size, amplified up to ~1e4 at a beam resonance)")
