"""pydisort_amd.subroutines (host helper library) against values computed by the reference's
PythonicDISORT.subroutines on the same inputs (tests/golden/helpers.npz, made by tests/golden/make_helper_goldens.py)."""
import os
import warnings

import numpy as np
import pytest

import goldens
from pydisort_amd import subroutines as S

Z = np.load(os.path.join(goldens.HERE, "golden", "helpers.npz"))
bdrf0 = lambda mu, nmup: 0.3 * (1 + 0.5 * np.outer(mu, nmup))  # noqa: E731


def close(a, b, tol=1e-13):
    return np.allclose(a, b, rtol=tol, atol=tol * max(1.0, float(np.max(np.abs(b)))))


def test_quadratures():
    for n, args in ((9, ()), (33, (-1.5, 2.0))):
        x, w = S.Clenshaw_Curtis_quad(n, *args)
        assert close(x, Z[f"cc{n}_x"]) and close(w, Z[f"cc{n}_w"])
    x, w = S.Gauss_Legendre_quad(7, -2, 3)
    assert close(x, Z["gl7_x"]) and close(w, Z["gl7_w"])
    with pytest.raises(ValueError):
        S.Clenshaw_Curtis_quad(8)


def test_finite_difference_matrix():
    g, D = S.generate_FD_mat(11, 0.5, 3.0)
    assert close(g, Z["fd_grid"]) and close(D.toarray(), Z["fd_mat"])


def test_planck_and_thermal_inputs():
    assert close(S.Planck(np.array([0.0, 200.0, 288.0, 320.0]), 60000.0), Z["planck"])
    assert close(S.blackbody_contrib_to_BCs(np.array([250.0, 300.0]), 30000.0, 120000.0), Z["bb"], 1e-10)
    assert close(S.blackbody_contrib_to_BCs(288.0, 0.0, 50000.0), Z["bb_scalar"], 1e-10)
    assert close(S.linear_spline_coefficients(np.array([0.0, 0.5, 2.0, 2.5]), np.array([1.0, 3.0, 2.0, 5.0])), Z["spline"])
    assert close(S.generate_s_poly_coeffs(np.array([0.3, 1.0, 4.0]), np.array([220.0, 250.0, 270.0, 295.0]), 30000.0, 80000.0),
                 Z["spoly"], 1e-10)
    with pytest.raises(ValueError):
        S.generate_s_poly_coeffs(np.array([0.3, 1.0]), np.array([220.0, 250.0]), 1.0, 2.0)


def test_bdrf_helpers():
    assert close(S.generate_emissivity_from_BDRF(8, 0.25), Z["emis_scalar"])
    assert close(S.generate_emissivity_from_BDRF(8, bdrf0), Z["emis_fn"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c1 = S.cache_BDRF_Fourier_modes(4, [0.2, bdrf0], mu0=0.6)
        c2 = S.cache_BDRF_Fourier_modes(4, [0.2, bdrf0])
    mu4 = S.Gauss_Legendre_quad(4)[0]
    assert close(c1[1](mu4, mu4), Z["cache_mu0_full"]) and close(c1[1](mu4, np.array([0.6])), Z["cache_mu0_col"])
    assert close(c1[0](mu4, mu4), Z["cache_scalar"])
    assert close(c2[1](mu4, mu4), Z["cache_nomu0_full"]) and close(c2[1](mu4, np.array([0.45])), Z["cache_nomu0_col"])


def test_small_utilities():
    assert close(S.affine_transform_poly_coeffs(np.array([[1.0, 2.0, 3.0], [0.5, -1.0, 4.0]]), np.array([0.8, 1.3]),
                                                np.array([0.1, -0.4])), Z["affine"])
    A = np.arange(36.0).reshape(6, 6) + 1
    assert np.array_equal(S.to_diag_ordered_form(A, 2, 1), Z["dof"])
    assert close(S.calculate_nu(np.array([0.2, -0.7]), np.array([0.0, 1.0, 2.0]), np.array([0.5]), np.array([0.3])), Z["nu"])
    assert np.array_equal(S.atleast_2d_append(np.arange(3.0)), Z["a2d"])
    assert np.array_equal(S.prepend(np.array([1.0, 2.0]), 2, 7.0), Z["prepend"])


@pytest.mark.gpu
def test_interpolate_and_actinic_flux_on_device_solution():
    """mu-interpolation and actinic fluxes built from the device closures: they reproduce the quadrature values at
    the nodes and obey the obvious integral identity with the energetic flux weights."""
    import pydisort_amd
    kw = goldens.load("9c")[0]["kwargs"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu_arr, Fp, Fm, u0, u = pydisort_amd.pydisort(**kw)
    ui = S.interpolate(u)
    tau, phi = np.array([0.5, 7.0]), np.array([0.0, 2.0])
    assert np.allclose(ui(mu_arr, tau, phi), u(tau, phi), rtol=1e-10, atol=1e-12)
    u0i = S.interpolate(u0)
    assert np.allclose(u0i(mu_arr, tau), u0(tau), rtol=1e-10, atol=1e-12)
    fa_up, fa_dn = S.generate_diff_act_flux_funcs(u0)
    N = len(mu_arr) // 2
    w = S.Gauss_Legendre_quad(N)[1]
    assert np.allclose(fa_up(tau), 2 * np.pi * w @ u0(tau)[:N])
    assert np.all(np.isfinite(fa_dn(tau)))


def test_hapke_samples_reproduce_reference_bdrf_tables():
    """SURVEY 8(f) f4, host half: Hapke_BDRF + sample_BDRF + the trapezoid cosine sum (what the device kernel computes)
    against the Fourier-mode tables the reference produced with scipy's quad_vec for test 6d (captured in the golden
    file).  Off the opposition cusp the rule is spectrally accurate; on it (mu = mu', dphi = pi) it is second order."""
    import numpy as np
    import goldens
    from pydisort_amd import subroutines as sub
    call = goldens.load("6d")[0]
    kw = call["kwargs"]
    nphi = 4096
    rho_qq, rho_q0 = sub.sample_BDRF(sub.Hapke_BDRF(1.0, 0.06, 0.6), kw["NQuad"], float(kw["mu0"]), nphi=nphi)
    p = np.arange(nphi)
    for m, f in enumerate(kw["BDRF_Fourier_modes"]):
        c = (1 if m == 0 else 2) / nphi * np.cos(2 * np.pi * m * p / nphi)
        q, q0 = rho_qq @ c, rho_q0 @ c
        off = ~np.eye(q.shape[0], dtype=bool)
        assert np.max(np.abs(q - f.tab)[off]) < 1e-9      # the reference's quad_vec tolerance
        assert np.max(np.abs(q - f.tab)) < 1e-5           # cusp of the opposition surge on the diagonal
        assert np.max(np.abs(q0 - f.tab0)) < 1e-9
