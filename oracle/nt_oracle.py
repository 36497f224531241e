"""CPU ORACLE (test infrastructure, NOT product code): Nakajima-Tanaka intensity corrections.

Restates the TMS and IMS corrections that the reference applies inside ``u_corrected``
(pydisort.py:375-698) on top of the delta-M scaled solution of oracle/disort_oracle.py.
Written with explicit per-layer sums instead of the reference's cumulative-sum formulation;
the quantities are the same (file:line cited per block).  Pinned by the same golden vectors as
the rest of the oracle (tests/test_oracle_vs_reference_goldens.py: TP3, 4, 5, 7c-d, 9corrections,
11a, Ia-c).
"""
import numpy as np
from numpy.polynomial.legendre import legval


def nt_active(p):
    """pydisort.py:375."""
    return bool(p["NT_cor"]) and not p["only_flux"] and p["beam"] and np.any(p["f"] > 0) \
        and p["P"] < p["NLeg_all"] and np.any(p["omega"] > 0)


def _nu(mu, phi, mu0, phi0):
    """Cosine of the scattering angle between (mu, phi) and the beam (-mu0, phi0) (subroutines.py:85-112)."""
    return (-mu0 * mu)[:, None] + (np.sqrt(1 - mu0**2) * np.sqrt(1 - mu**2))[:, None] * np.cos(phi0 - phi)[None, :]


def tms(sol, tau, l, ts, phi, antider):
    """TMS correction [Q, Ntau, Nphi] (pydisort.py:409-596)."""
    p = sol.p
    N, L, mu, mu0 = p["N"], p["L"], p["mu"], p["mu0"]
    ts0, dts, sc = p["tau_s0"], p["thick_s"], p["scale_tau"]
    mu_all = sol.mu_arr
    nu = _nu(mu_all, phi, mu0, p["phi0"])  # [Q, Nphi]
    wfull = p["Leg_all"] * (2 * np.arange(p["NLeg_all"]) + 1)[None, :]
    # mathscr_B for every layer: [Q, L, Nphi]  (:424-449)
    calB = np.empty((p["Q"], L, len(phi)))
    for r in range(L):
        p_true = legval(nu, wfull[r])
        p_trun = legval(nu, p["wleg"][r])
        calB[:, r, :] = (p["omega_s"][r] * p["I0_4pi"]) * (mu0 / (mu0 + mu_all))[:, None] \
            * (p_true / (1 - p["f"][r]) - p_trun)
    tb, tt = ts0[1:][l], ts0[l]  # scaled tau at the bottom / top of the point's layer
    beam_att = np.exp(-ts / mu0)
    if antider:  # :455-470
        c0 = beam_att / (-sc[l] / mu0)
        up = c0[None, :] - np.exp((ts - tb)[None, :] / mu[:, None] - tb[None, :] / mu0) / (sc[l][None, :] / mu[:, None])
        dn = c0[None, :] + np.exp((tt - ts)[None, :] / mu[:, None] - tt[None, :] / mu0) / (sc[l][None, :] / mu[:, None])
    else:  # :471-479
        up = beam_att[None, :] - np.exp((ts - tb)[None, :] / mu[:, None] - tb[None, :] / mu0)
        dn = beam_att[None, :] - np.exp((tt - ts)[None, :] / mu[:, None] - tt[None, :] / mu0)
    inlayer = np.concatenate((up, dn), axis=0)  # [Q, Ntau]
    if L > 1:  # contributions of the other layers (:489-589); note the reference weights them with
        # mathscr_B of the layer CONTAINING the point, reproduced here.
        Rpos = np.zeros((N, L))
        Rneg = np.zeros((N, L))
        intf = (mu[:, None] / sc[None, :]) if antider else np.ones((N, L))
        for r in range(L):
            # upward streams: layer r below the point's layer  (:507-530)
            tpos = (1 - np.exp(-dts[r] * (1 / mu + 1 / mu0))) * intf[:, r] * np.exp(-ts0[r] / mu0)
            for ll in range(r):
                Rpos[:, ll] += tpos * np.exp(-(ts0[r] - ts0[ll + 1]) / mu)
            # downward streams: layer r above the point's layer  (:543-575)
            d = dts[r] * (1 / mu - 1 / mu0)
            em1 = np.expm1(-np.abs(d))
            tneg = np.where(d >= 0, -em1 * np.exp(-ts0[r + 1] / mu0),
                            em1 * np.exp(-dts[r] / mu) * np.exp(-ts0[r] / mu0))
            if antider:
                tneg = -intf[:, r] * tneg
            for ll in range(r + 1, L):
                Rneg[:, ll] += tneg * np.exp(-(ts0[ll] - ts0[r + 1]) / mu)
        inlayer[:N] += Rpos[:, l] * np.exp((ts - tb)[None, :] / mu[:, None])
        inlayer[N:] += Rneg[:, l] * np.exp((tt - ts)[None, :] / mu[:, None])
    return calB[:, l, :] * inlayer[:, :, None]


def ims(sol, tau, phi, antider):
    """IMS correction for the downward streams [N, Ntau, Nphi] (pydisort.py:601-638)."""
    p = sol.p
    mu, mu0 = p["mu"], p["mu0"]
    w = p["omega"] * p["tau"]
    omega_avg = w.sum() / p["tau"].sum()
    f_avg = (p["f"] * w).sum() / w.sum()
    resid = p["Leg_all"].copy()
    resid[:, :p["P"]] = p["f"][:, None]
    resid_avg = (resid * w[:, None]).sum(axis=0) / (p["f"] * w).sum()
    smu0 = mu0 / (1 - omega_avg * f_avg)
    nu = _nu(-mu, phi, mu0, p["phi0"])
    x = 1 / mu - 1 / smu0
    if antider:
        chi = ((smu0 - x[:, None] * smu0 * (smu0 + tau)[None, :]) * np.exp(-tau / smu0)[None, :]
               - mu[:, None] * np.exp(-tau[None, :] / mu[:, None])) / (mu * smu0 * x**2)[:, None]
    else:
        chi = ((tau[None, :] - 1 / x[:, None]) * np.exp(-tau / smu0)[None, :]
               + np.exp(-tau[None, :] / mu[:, None]) / x[:, None]) / (mu * smu0 * x)[:, None]
    series = legval(nu, (2 * np.arange(p["NLeg_all"]) + 1) * (2 * resid_avg - resid_avg**2))
    amp = p["I0_4pi"] * (omega_avg * f_avg) ** 2 / (1 - omega_avg * f_avg)
    return (amp * series)[:, None, :] * chi[:, :, None]


def corrected_u(sol):
    """The closure the reference returns as ``u`` when the corrections are active (:643-694)."""
    p = sol.p

    def u_corrected(tau, phi, is_antiderivative_wrt_tau=False, return_Fourier_error=False, return_tau_arr=False):
        tau_a = np.atleast_1d(np.asarray(tau, dtype=float))
        phi_a = np.atleast_1d(np.asarray(phi, dtype=float))
        base = sol.u(tau_a, phi_a, is_antiderivative_wrt_tau, return_Fourier_error, return_tau_arr)
        _, l, ts = sol._locate(tau_a)
        corr = tms(sol, tau_a, l, ts, phi_a, is_antiderivative_wrt_tau)
        corr[p["N"]:] += ims(sol, tau_a, phi_a, is_antiderivative_wrt_tau)
        corr = p["rescale"] * np.squeeze(corr)
        if isinstance(base, tuple):
            return (base[0] + corr,) + base[1:]
        return base + corr

    return u_corrected
