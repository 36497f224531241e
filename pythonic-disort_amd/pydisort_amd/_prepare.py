"""Host-side preparation of the hot path's arguments for a batch of columns.

Mirror of the front end of the reference (src/PythonicDISORT/pydisort.py:184-372): double-Gauss
quadrature (:304-306), delta-M scaling (:316-338), source rescaling (:351-372) -- vectorised over a
leading column axis C.  Everything after this (eigen stage, boundary-condition solve, evaluation)
runs on the GPU.
"""
from functools import lru_cache
from math import comb

import numpy as np


@lru_cache(maxsize=64)
def _double_gauss_cached(N):
    x, w = np.polynomial.legendre.leggauss(N)
    mu, W = 0.5 * (x + 1.0), 0.5 * w
    mu.setflags(write=False)
    W.setflags(write=False)
    return mu, W


def double_gauss(N):
    """Gauss-Legendre nodes and weights on [0, 1] (pydisort.py:304; subroutines.py:116-138).  Cached per N (the eigenvalue
    problem behind `leggauss` was 40 % of a one-column call's latency); the cached arrays are read-only."""
    return _double_gauss_cached(int(N))


def _recentre_poly(coef, a, b):
    """Given p(x) = sum_j coef[..., j] x^j and y = a x + b, coefficients of p as a polynomial in y
    (what subroutines.affine_transform_poly_coeffs computes, subroutines.py:574-610)."""
    n = coef.shape[-1]
    out = np.zeros_like(coef)
    for jj in range(n):
        for i in range(jj + 1):
            out[..., i] += coef[..., jj] * comb(jj, i) * (-b) ** (jj - i) / a**jj
    return out


def prepare_columns(tau_arr, omega_arr, NQuad, Leg_coeffs_all, mu0, I0, phi0, NLeg, NFourier, b_pos, b_neg,
                    f_arr, s_poly_coeffs, bdrf_q, bdrf_q0):
    """All arguments carry a leading column axis:
    tau_arr, omega_arr, f_arr [C, L]; Leg_coeffs_all [C, L, NLeg_all]; mu0, I0, phi0 [C];
    b_pos, b_neg [C, N, NFourier]; s_poly_coeffs [C, L, Ns] (Ns may be 0);
    bdrf_q [C, NBDRF, N, N], bdrf_q0 [C, NBDRF, N] (NBDRF may be 0).
    Returns the dict consumed by _engine.Plan."""
    tau_arr = np.asarray(tau_arr, float)
    C, L = tau_arr.shape
    N = NQuad // 2
    omega = np.asarray(omega_arr, float)
    f = np.broadcast_to(np.asarray(f_arr, float), (C, L))
    leg = np.asarray(Leg_coeffs_all, float)[:, :, :NLeg]
    s_poly = np.asarray(s_poly_coeffs, float).reshape(C, L, -1)
    Ns = s_poly.shape[2]
    mu, W = double_gauss(N)
    thick = np.diff(tau_arr, axis=1, prepend=0.0)
    ell = np.arange(NLeg)
    if np.any(f > 0):  # delta-M scaling (:316-329)
        scale_tau = 1.0 - omega * f
        tau_s0 = np.concatenate((np.zeros((C, 1)), np.cumsum(scale_tau * thick, axis=1)), axis=1)
        leg_s = (leg - f[:, :, None]) / (1.0 - f)[:, :, None]
        omega_s = (1.0 - f) / scale_tau * omega
        if Ns > 0:
            top = np.concatenate((np.zeros((C, 1)), tau_arr[:, :-1]), axis=1)
            shift = tau_s0[:, :-1] - scale_tau * top
            s_s = _recentre_poly(s_poly, scale_tau, shift) \
                / scale_tau[:, :, None] * (1.0 - omega)[:, :, None]
        else:
            s_s = np.zeros((C, L, 0))
    else:  # (:331-338)
        scale_tau = np.ones((C, L))
        tau_s0 = np.concatenate((np.zeros((C, 1)), tau_arr), axis=1)
        leg_s = leg
        omega_s = omega.copy()
        s_s = s_poly * (1.0 - omega)[:, :, None] if Ns > 0 else np.zeros((C, L, 0))
    wleg = leg_s * (2 * ell + 1)[None, None, :]

    I0 = np.asarray(I0, float).reshape(C)
    # b_pos / b_neg = None: all zero (nothing is allocated or uploaded for them)
    b_pos = None if b_pos is None else np.asarray(b_pos, float).reshape(C, N, NFourier)
    b_neg = None if b_neg is None else np.asarray(b_neg, float).reshape(C, N, NFourier)
    # rescale of the sources (:351-372): max(I0, max b_pos, max b_neg[, s(0) top, s(tau_L) bottom])
    cand = [I0, np.zeros(C) if b_pos is None else b_pos.reshape(C, -1).max(axis=1),
            np.zeros(C) if b_neg is None else b_neg.reshape(C, -1).max(axis=1)]
    if Ns > 0:
        cand.append(s_s[:, 0, 0])
        cand.append(np.einsum("cj,cj->c", s_s[:, -1, :], tau_s0[:, -1:] ** np.arange(Ns)[None, :]))
    rescale = np.max(np.stack(cand, axis=0), axis=0)
    div = np.where(rescale != 0, rescale, 1.0) if Ns == 0 else rescale
    I0s = I0 / div
    b_pos = None if b_pos is None else b_pos / div[:, None, None]
    b_neg = None if b_neg is None else b_neg / div[:, None, None]
    if Ns > 0:
        s_s = s_s / div[:, None, None]
    if Ns == 0:
        rescale = np.where(rescale != 0, rescale, 0.0)
    bdrf_q = np.asarray(bdrf_q, float).reshape(C, -1, N, N)
    bdrf_q0 = np.asarray(bdrf_q0, float).reshape(C, -1, N)
    return dict(C=C, L=L, N=N, P=NLeg, M=NFourier, Ns=Ns, NBDRF=bdrf_q.shape[1], beam=bool(np.any(I0 > 0)),
                mu=mu, W=W, omega_s=omega_s, tau=tau_arr, tau_s0=tau_s0, scale_tau=scale_tau, wleg=wleg,
                mu0=np.asarray(mu0, float).reshape(C), I0=I0s, phi0=np.asarray(phi0, float).reshape(C),
                rescale=np.asarray(rescale, float),
                b_pos=None if b_pos is None else np.ascontiguousarray(b_pos.transpose(0, 2, 1)),  # -> [C, M, N]
                b_neg=None if b_neg is None else np.ascontiguousarray(b_neg.transpose(0, 2, 1)),
                s_s=s_s if Ns > 0 else None,
                bdrf_q=bdrf_q if bdrf_q.shape[1] > 0 else None,
                bdrf_q0=bdrf_q0 if bdrf_q.shape[1] > 0 else None,
                leg_s=leg_s)
