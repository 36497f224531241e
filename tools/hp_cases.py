"""Atmospheres of the high-precision accuracy check (tools/hp_truth_m0.py, tools/hp_compare.py)."""
import numpy as np


def harsh_case():
    L, NQuad = 6, 16
    tau = np.cumsum([0.5, 2.0, 0.05, 8.0, 1.0, 3.0])
    omega = np.array([0.9, 1 - 1e-6, 0.5, 1 - 1e-6, 0.99, 0.2])
    g = np.array([0.7, 0.85, 0.3, 0.8, 0.6, 0.1])
    Leg = g[:, None] ** np.arange(NQuad + 4)[None, :]
    return dict(tau_arr=tau, omega_arr=omega, NQuad=NQuad, Leg_coeffs_all=Leg, mu0=0.69, I0=2.0, phi0=0.0,
                f_arr=g**NQuad, b_pos=0.3, b_neg=0.1, only_flux=True)


def benign_case():
    kw = harsh_case()
    kw["omega_arr"] = np.array([0.9, 0.95, 0.5, 0.8, 0.99, 0.2])
    return kw


def intensity_case():
    """Multi-mode check: 8 streams, 8 Fourier modes, strongly forward-peaked layers, one omega = 1 - 1e-6 layer."""
    L, NQuad = 4, 8
    tau = np.cumsum([0.3, 4.0, 0.7, 2.0])
    omega = np.array([0.95, 1 - 1e-6, 0.6, 0.9])
    g = np.array([0.8, 0.75, 0.4, 0.65])
    Leg = g[:, None] ** np.arange(NQuad + 3)[None, :]
    return dict(tau_arr=tau, omega_arr=omega, NQuad=NQuad, Leg_coeffs_all=Leg, mu0=0.55, I0=3.0, phi0=0.4,
                f_arr=g**NQuad, b_pos=0.2, b_neg=0.05)


PHI = np.array([0.0, 1.3, 3.0])
