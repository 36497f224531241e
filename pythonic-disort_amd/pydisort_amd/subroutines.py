"""Small host helpers mirrored from PythonicDISORT.subroutines that the drop-in path itself needs
(only Gauss_Legendre_quad, calculate_nu and friends are on the path; the Planck /
BDRF-cache / interpolation helpers are row f3, "next")."""
import numpy as np

from ._prepare import double_gauss


def Gauss_Legendre_quad(N, c=0, d=1):
    """Gauss-Legendre nodes and weights for integration over [c, d] (subroutines.py:116-138)."""
    x, w = np.polynomial.legendre.leggauss(int(N))
    return (x + 1) * (d - c) / 2 + c, w * (d - c) / 2


def calculate_nu(mu, phi, mu_p, phi_p):
    """Cosine of the scattering angle between (mu_p, phi_p) and (mu, phi); axes (mu, phi, mu_p, phi_p),
    squeezed (subroutines.py:85-112)."""
    mu, phi, mu_p, phi_p = np.atleast_1d(mu, phi, mu_p, phi_p)
    nu = mu_p[None, None, :, None] * mu[:, None, None, None] + np.sqrt(1 - mu_p**2)[None, None, :, None] \
        * np.sqrt(1 - mu**2)[:, None, None, None] * np.cos(phi_p[None, None, None, :] - phi[None, :, None, None])
    return np.squeeze(nu)


def compare_with_stamnes(results, mu_to_compare, reorder_mu, flux_up, flux_down, u=None):
    """Pointwise abs / relative differences against stored DISORT 4.0.99 results, in the order the
    reference's ``_compare`` returns them (subroutines.py:866-975)."""
    tau = results["tau_test_arr"]

    def pair(ref, got):
        diff = np.abs(ref - got)
        ratio = np.divide(diff, ref, out=np.zeros_like(diff), where=(ref != 0))
        return diff, ratio

    out = pair(results["flup"], flux_up(tau))
    fd = flux_down(tau)
    out += pair(results["rfldn"], fd[0]) + pair(results["rfldir"], fd[1])
    if u is not None:
        uu = results["uu"]
        got = u(tau, results["phi_arr"])[reorder_mu].reshape(uu.shape)
        diff = np.abs(uu - got)[mu_to_compare]
        ratio = np.divide(diff, uu[mu_to_compare], out=np.zeros_like(diff), where=(uu[mu_to_compare] != 0))
        out += (diff, ratio)
    return out


__all__ = ["Gauss_Legendre_quad", "calculate_nu", "compare_with_stamnes", "double_gauss"]
