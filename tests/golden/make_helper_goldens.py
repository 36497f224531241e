#!/usr/bin/env python3
"""Outputs of the REFERENCE's helper functions (PythonicDISORT.subroutines) on fixed inputs (this container only).
Pins pydisort_amd.subroutines.  Usage: PYTHONDONTWRITEBYTECODE=1 python3 <this file>"""
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
from PythonicDISORT import subroutines as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
x, w = R.Clenshaw_Curtis_quad(9)
out["cc9_x"], out["cc9_w"] = x, w
x, w = R.Clenshaw_Curtis_quad(33, -1.5, 2.0)
out["cc33_x"], out["cc33_w"] = x, w
x, w = R.Gauss_Legendre_quad(7, -2, 3)
out["gl7_x"], out["gl7_w"] = x, w
g, D = R.generate_FD_mat(11, 0.5, 3.0)
out["fd_grid"], out["fd_mat"] = g, D.toarray()
out["planck"] = R.Planck(np.array([0.0, 200.0, 288.0, 320.0]), 60000.0)
out["bb"] = R.blackbody_contrib_to_BCs(np.array([250.0, 300.0]), 30000.0, 120000.0)
out["bb_scalar"] = np.array(R.blackbody_contrib_to_BCs(288.0, 0.0, 50000.0))
out["spline"] = R.linear_spline_coefficients(np.array([0.0, 0.5, 2.0, 2.5]), np.array([1.0, 3.0, 2.0, 5.0]))
out["spoly"] = R.generate_s_poly_coeffs(np.array([0.3, 1.0, 4.0]), np.array([220.0, 250.0, 270.0, 295.0]), 30000.0, 80000.0)
bdrf0 = lambda mu, nmup: 0.3 * (1 + 0.5 * np.outer(mu, nmup))
out["emis_scalar"] = np.array(R.generate_emissivity_from_BDRF(8, 0.25))
out["emis_fn"] = R.generate_emissivity_from_BDRF(8, bdrf0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    c1 = R.cache_BDRF_Fourier_modes(4, [0.2, bdrf0], mu0=0.6)
    c2 = R.cache_BDRF_Fourier_modes(4, [0.2, bdrf0])
mu4 = R.Gauss_Legendre_quad(4)[0]
out["cache_mu0_full"] = c1[1](mu4, mu4)
out["cache_mu0_col"] = c1[1](mu4, np.array([0.6]))
out["cache_scalar"] = np.array(c1[0](mu4, mu4))
out["cache_nomu0_full"] = c2[1](mu4, mu4)
out["cache_nomu0_col"] = c2[1](mu4, np.array([0.45]))
out["affine"] = R.affine_transform_poly_coeffs(np.array([[1.0, 2.0, 3.0], [0.5, -1.0, 4.0]]), np.array([0.8, 1.3]), np.array([0.1, -0.4]))
A = np.arange(36.0).reshape(6, 6) + 1
out["dof"] = R.to_diag_ordered_form(A, 2, 1)
out["nu"] = R.calculate_nu(np.array([0.2, -0.7]), np.array([0.0, 1.0, 2.0]), np.array([0.5]), np.array([0.3]))
out["a2d"] = R.atleast_2d_append(np.arange(3.0))
out["prepend"] = R.prepend(np.array([1.0, 2.0]), 2, 7.0)
np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)
print("wrote", len(out), "arrays")
