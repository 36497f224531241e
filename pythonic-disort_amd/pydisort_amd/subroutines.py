"""Host helper library mirroring ``PythonicDISORT.subroutines``.

Same names, argument meaning and return shapes as the reference's helpers (file:line cited per function,
relative to src/PythonicDISORT/subroutines.py) so that scripts written against the reference -- building thermal
sources, blackbody boundary terms, emissivities, cached BDRF modes, actinic fluxes, mu-interpolation -- run
unchanged on top of ``pydisort_amd.pydisort``.  These are input/output conveniences on the host; the solver itself
runs on the GPU.  Parity with the reference's outputs is pinned by tests/golden/helpers.npz.
"""
import inspect
import warnings
from math import comb, pi

import numpy as np
import scipy.constants
import scipy.integrate
import scipy.interpolate
import scipy.sparse

from ._prepare import double_gauss


def prepend(arr, arr_len, value):
    """Array of length ``arr_len + 1`` with ``value`` in front of ``arr`` (:7-29)."""
    out = np.empty(arr_len + 1)
    out[0] = value
    out[1:] = arr
    return out


def transform_interval(arr, c, d, a, b):
    """Affine map of points from [a, b] to [c, d] (:33-54)."""
    return c + (np.asarray(arr) - a) * ((d - c) / (b - a))


def transform_weights(weights, c, d, a, b):
    """Quadrature weights of [a, b] rescaled to [c, d] (:58-79)."""
    return np.asarray(weights) * ((d - c) / (b - a))


def calculate_nu(mu, phi, mu_p, phi_p):
    """Cosine of the scattering angle between (mu_p, phi_p) and (mu, phi); axes (mu, phi, mu_p, phi_p),
    squeezed (:83-112)."""
    mu, phi, mu_p, phi_p = np.atleast_1d(mu, phi, mu_p, phi_p)
    nu = mu_p[None, None, :, None] * mu[:, None, None, None] + np.sqrt(1 - mu_p**2)[None, None, :, None] \
        * np.sqrt(1 - mu**2)[:, None, None, None] * np.cos(phi_p[None, None, None, :] - phi[None, :, None, None])
    return np.squeeze(nu)


def Gauss_Legendre_quad(N, c=0, d=1):
    """Gauss-Legendre nodes and weights for integration over [c, d] (:116-138)."""
    x, w = np.polynomial.legendre.leggauss(int(N))
    return transform_interval(x, c, d, -1, 1), transform_weights(w, c, d, -1, 1)


def Clenshaw_Curtis_quad(Nphi, c=0, d=(2 * pi)):
    """Clenshaw-Curtis nodes (ascending) and weights for integration over [c, d]; Nphi odd and > 2 (:142-175)."""
    if not (Nphi > 2 and Nphi % 2 == 1):
        raise ValueError("The number of quadrature nodes must be odd and greater than 2.")
    n = Nphi - 1
    k = np.arange(Nphi)
    x = -np.cos(pi * k / n)
    x[n // 2] = 0.0
    jj = np.arange(1, n // 2 + 1)
    b = np.where(jj == n // 2, 1.0, 2.0)
    w = 1.0 - (b[None, :] / (4.0 * jj[None, :] ** 2 - 1.0) * np.cos(2.0 * jj[None, :] * k[:, None] * pi / n)).sum(axis=1)
    w *= np.where((k == 0) | (k == n), 1.0, 2.0) / n
    return transform_interval(x, c, d, -1, 1), transform_weights(w, c, d, -1, 1)


def generate_FD_mat(Ntau, a, b):
    """Grid on [a, b] and the sparse (csr) second-order first-derivative matrix: central differences inside,
    one-sided second-order stencils at both ends (:179-213)."""
    grid = np.linspace(a, b, Ntau)
    h = grid[1] - grid[0]
    D = scipy.sparse.lil_matrix((Ntau, Ntau))
    idx = np.arange(1, Ntau - 1)
    D[idx, idx + 1] = 1 / (2 * h)
    D[idx, idx - 1] = -1 / (2 * h)
    D[0, :3] = np.array([-3, 4, -1]) / (2 * h)
    D[-1, -3:] = np.array([1, -4, 3]) / (2 * h)
    return grid, D.tocsr()


def atleast_2d_append(*arys):
    """``numpy.atleast_2d`` that appends the new axes instead of prepending them (:217-254)."""
    out = []
    for a in arys:
        a = np.asanyarray(a)
        out.append(a.reshape(1, 1) if a.ndim == 0 else (a[:, None] if a.ndim == 1 else a))
    return out[0] if len(out) == 1 else out


def generate_diff_act_flux_funcs(u0):
    """Up and down diffuse actinic flux functions from the zeroth intensity mode ``u0`` returned by ``pydisort``,
    with the reclassification of delta-scaled actinic flux (:258-318)."""
    N = len(u0(0)) // 2
    wts = Gauss_Legendre_quad(N)[1]

    def flux_act_up(tau, is_antiderivative_wrt_tau=False, return_tau_arr=False):
        if return_tau_arr:
            val, tau_arr = u0(tau, is_antiderivative_wrt_tau, True)
            return np.squeeze(2 * pi * wts @ val[:N])[()], tau_arr
        return np.squeeze(2 * pi * wts @ u0(tau, is_antiderivative_wrt_tau)[:N])[()]

    def flux_act_down_diffuse(tau, is_antiderivative_wrt_tau=False, return_tau_arr=False):
        if return_tau_arr:
            val, tau_arr, reclass = u0(tau, is_antiderivative_wrt_tau, True, _return_act_dscale_for_reclass=True)
            return np.squeeze(2 * pi * wts @ val[N:] + reclass)[()], tau_arr
        val, reclass = u0(tau, is_antiderivative_wrt_tau, False, _return_act_dscale_for_reclass=True)
        return np.squeeze(2 * pi * wts @ val[N:] + reclass)[()]

    return flux_act_up, flux_act_down_diffuse


def Planck(T, WVNM):
    """Planck function for the intensity leaving a blackbody surface in W / m^2 per unit wavenumber, the units of
    Stamnes' DISORT; T in kelvin (0 gives 0), WVNM in m^-1 (:322-350)."""
    T = np.atleast_1d(np.asarray(T, dtype=float))
    out = np.zeros(len(T))
    ok = T != 0
    if np.any(ok):
        h, c, k = scipy.constants.h, scipy.constants.c, scipy.constants.k
        e = np.exp(-100 * h * c * WVNM / (k * T[ok]))  # written with exp(-x) so that it cannot overflow
        out[ok] = 2e8 * h * c**2 * WVNM**3 * e / (1 - e)
    return np.squeeze(out)[()]


def blackbody_contrib_to_BCs(T, WVNMLO, WVNMHI, **kwargs):
    """Blackbody emission of a boundary at temperature(s) T integrated over [WVNMLO, WVNMHI] (m^-1), W / m^2;
    ``kwargs`` go to ``scipy.integrate.quad_vec`` (:354-377)."""
    return np.squeeze(scipy.integrate.quad_vec(lambda v: Planck(T, v), WVNMLO, WVNMHI, **kwargs)[0])


def linear_spline_coefficients(x, y, check_inputs=True):
    """Per-segment (intercept, slope) of the linear spline through (x, y): axes (segment, ascending order) (:381-409)."""
    x, y = np.asarray(x, float), np.asarray(y, float)
    if check_inputs:
        if not len(x) > 1:
            raise ValueError("At least 2 points are required.")
        if not len(x) == len(y):
            raise ValueError("The number of x and y points must be equal.")
        if not np.all(np.diff(x) > 0):
            raise ValueError("The x values must be sorted in ascending order.")
    slope = np.diff(y) / np.diff(x)
    return np.stack((y[:-1] - slope * x[:-1], slope), axis=1)


def generate_s_poly_coeffs(tau_arr, TEMPER, WVNMLO, WVNMHI, **kwargs):
    """DISORT-equivalent ``s_poly_coeffs``: the Planck emission integrated over the wavenumber band at every
    level, interpolated linearly in tau inside each layer (:413-454)."""
    tau_arr = np.atleast_1d(tau_arr)
    if not len(TEMPER) == len(tau_arr) + 1:
        raise ValueError("Missing temperature specification at some boundaries / interfaces.")
    levels = prepend(tau_arr, len(tau_arr), 0)
    emission = scipy.integrate.quad_vec(lambda v: Planck(TEMPER, v), WVNMLO, WVNMHI, **kwargs)[0]
    return linear_spline_coefficients(levels, emission, check_inputs=False)


def generate_emissivity_from_BDRF(N, zeroth_BDRF_Fourier_mode):
    """Directional emissivity of the surface by Kirchhoff's law from the zeroth BDRF Fourier mode (scalar or
    callable f(mu, -mu')) (:459-486)."""
    if np.isscalar(zeroth_BDRF_Fourier_mode):
        return 1 - zeroth_BDRF_Fourier_mode
    mu, w = Gauss_Legendre_quad(N)
    return 1 - 2 * (zeroth_BDRF_Fourier_mode(mu, mu) * mu[None, :]) @ w


def cache_BDRF_Fourier_modes(N, BDRF_Fourier_modes, mu0=0):
    """BDRF Fourier modes evaluated once on the quadrature grid (and at ``mu0`` when 0 < mu0 <= 1) and wrapped as
    callables with the signature ``pydisort`` expects (:490-570)."""
    with_mu0 = 0 < mu0 <= 1
    if not with_mu0:
        warnings.warn("No caching with respect to `mu0`.")
    mu = Gauss_Legendre_quad(N)[0]
    cached = []
    for f in BDRF_Fourier_modes:
        if np.isscalar(f):
            cached.append(lambda mu_, neg_mup, v=f: v)
        elif with_mu0:
            tab = f(mu, np.append(mu, mu0))
            cached.append(lambda mu_, neg_mup, t=tab: t[:, [-1]] if len(neg_mup) == 1 else t[:, :-1])
        else:
            tab = f(mu, mu)
            cached.append(lambda mu_, neg_mup, t=tab, g=f: g(mu, neg_mup) if len(neg_mup) == 1 else t)
    return cached


def sample_BDRF(BDRF, NQuad, mu0=None, nphi=256):
    """Samples of a reflectance ``BDRF(mu, mu_p, dphi)`` (broadcasting over arrays mu [N,1], mu_p [1,N] and a scalar
    dphi) for the device-side Fourier integration (``Plan.set_bdrf_samples`` / ``pydisort_batch(bdrf_samples=...)``):
    returns (rho_qq [N, N, nphi], rho_q0 [N, nphi] or None) at the double-Gauss nodes of ``NQuad`` streams and
    dphi_p = 2 pi p / nphi.  The device forms q^m = 1/((1 + delta_m0) pi) Int_0^2pi BDRF cos(m dphi) ddphi by the
    trapezoid rule; the reference's tests integrate each mode on the host instead (pydisotest/6_test.py:194-201)."""
    N = NQuad // 2
    mu, _ = double_gauss(N)
    rho_qq = np.empty((N, N, nphi))
    rho_q0 = None if mu0 is None else np.empty((N, nphi))
    for p in range(nphi):
        dphi = 2 * pi * p / nphi
        rho_qq[:, :, p] = BDRF(mu[:, None], mu[None, :], dphi)
        if mu0 is not None:
            rho_q0[:, p] = np.asarray(BDRF(mu[:, None], np.atleast_1d(float(mu0))[None, :], dphi))[:, 0]
    return rho_qq, rho_q0


def Hapke_BDRF(B0, HH, W):
    """Hapke's reflectance as used by DISORT's test problem 6 (cf. pydisotest/6_test.py:11-25): returns
    rho(mu, mu_p, dphi) with the opposition surge B0 HH / (HH + tan(alpha/2)), the Legendre phase term 1 + cos(alpha)/2
    and the two-stream H functions; alpha is the phase angle between the incident and the reflected directions."""
    gamma = np.sqrt(1 - W)

    def rho(mu, mu_p, dphi):
        mu, mu_p = np.asarray(mu, float), np.asarray(mu_p, float)
        cos_a = np.clip(mu * mu_p - np.sqrt(1 - mu**2) * np.sqrt(1 - mu_p**2) * np.cos(dphi), -1.0, 1.0)
        surge = B0 * HH / (HH + np.tan(np.arccos(cos_a) / 2))
        h = (1 + 2 * mu) / (1 + 2 * mu * gamma) * (1 + 2 * mu_p) / (1 + 2 * mu_p * gamma)
        return W / 4 / (mu + mu_p) * ((1 + surge) * (1 + cos_a / 2) + h - 1)

    return rho


def affine_transform_poly_coeffs(poly_coeffs, a_arr, b_arr):
    """Rows of coefficients [C_0..C_n] of C(x); returns, per row, the coefficients [D_0..D_n] of the same function
    written in y = a x + b (:574-610)."""
    a_arr, b_arr = np.asarray(a_arr, float), np.asarray(b_arr, float)
    if np.any(a_arr) == 0:
        raise ValueError("The scale factors must be non-zero.")
    coef = np.atleast_2d(np.asarray(poly_coeffs, float))
    n = coef.shape[1]
    out = np.zeros_like(coef)
    for jj in range(n):
        for i in range(jj + 1):
            out[:, i] += coef[:, jj] * comb(jj, i) * (-b_arr) ** (jj - i) / a_arr**jj
    return out


def interpolate(u):
    """Barycentric polynomial interpolation in mu of ``u`` (tau, phi) or ``u0`` (tau) as returned by ``pydisort``:
    returns ``u_interpol(mu, tau[, phi], ...)`` continuous in every argument; the two hemispheres are interpolated
    separately (:614-705)."""
    params = list(inspect.signature(u).parameters)
    takes_phi = "phi" in params
    if not takes_phi and "tau" not in params:
        raise ValueError("This subroutine can only interpolate u or u0.")
    N = len(u(0, 0) if takes_phi else u(0)) // 2
    mu_pos = Gauss_Legendre_quad(N)[0]
    up = scipy.interpolate.BarycentricInterpolator(mu_pos)
    dn = scipy.interpolate.BarycentricInterpolator(-mu_pos)

    def _interp(mu, base, extra):
        results = np.empty((len(mu),) + np.shape(base)[1:])
        pos = mu > 0
        if np.any(pos):
            up.set_yi(base[:N])
            results[pos] = up(mu[pos])
        if np.any(~pos):
            dn.set_yi(base[N:])
            results[~pos] = dn(mu[~pos])
        val = np.squeeze(results)[()]
        return (val,) + extra if extra is not None else val

    if takes_phi:
        def u_interpol(mu, tau, phi, is_antiderivative_wrt_tau=False, return_Fourier_error=False, return_tau_arr=False):
            if not np.all(np.abs(mu) <= 1):
                raise ValueError("mu values must be between -1 and 1.")
            mu = np.atleast_1d(mu)
            if return_Fourier_error or return_tau_arr:
                outs = u(tau, phi, is_antiderivative_wrt_tau, return_Fourier_error, return_tau_arr)
                return _interp(mu, outs[0], outs[1:])
            return _interp(mu, u(tau, phi, is_antiderivative_wrt_tau), None)
    else:
        def u_interpol(mu, tau, is_antiderivative_wrt_tau=False, return_tau_arr=False):
            if not np.all(np.abs(mu) <= 1):
                raise ValueError("mu values must be between -1 and 1.")
            mu = np.atleast_1d(mu)
            if return_tau_arr:
                outs = u(tau, is_antiderivative_wrt_tau, return_tau_arr)
                return _interp(mu, outs[0], outs[1:])
            return _interp(mu, u(tau, is_antiderivative_wrt_tau), None)
    return u_interpol


def to_diag_ordered_form(A, Nsuperdiags, Nsubdiags):
    """Square matrix -> the diagonal ordered form of ``scipy.linalg.solve_banded``, as the reference's helper builds
    it (wrap-around entries outside the band are kept, exactly like the reference's fancy indexing) (:709-742)."""
    A = np.asarray(A)
    n = A.shape[0]
    cols = np.arange(n)
    rows = np.concatenate((np.arange(Nsuperdiags, -1, -1), np.arange(n - 1, n - Nsubdiags - 1, -1)))
    return A[(cols[None, :] - rows[:, None]), cols[None, :]]


def compare_with_stamnes(results, mu_to_compare, reorder_mu, flux_up, flux_down, u=None):
    """Pointwise abs / relative differences against stored DISORT 4.0.99 results, in the order the reference's
    ``_compare`` returns them (:866-975)."""
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


_compare = compare_with_stamnes
