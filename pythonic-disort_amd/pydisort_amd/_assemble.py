"""The reference's inner boundary, executable: ``_assemble_intensity_and_fluxes`` with the reference's exact positional
signature (src/PythonicDISORT/_assemble_intensity_and_fluxes.py:8-32), built on the C ABI of include/rtd.h.

This is the function a maintainer of PythonicDISORT would bind instead of their own: ``pydisort.py:381-405`` and ``:701-725``
call it with the 34 prepared arguments (delta-M scaled optical properties, quadrature, rescaled sources) and get back the
solution callables ``(flux_up, flux_down, u0[, u])`` (``:616-619``).  Here the arguments are laid out as one column of a
device plan (``rtd_plan_set_quadrature`` / ``rtd_plan_set_columns``), ``rtd_plan_solve`` replaces
``_solve_for_gen_and_part_sols`` + ``_solve_for_coeffs`` (``:109-159``), and the callables evaluate through
``rtd_plan_evaluate`` (``:170-613``).  ``tests/test_gpu_assemble_shim.py`` replays positional arguments captured from the
reference itself (tests/golden/assemble/*.npz, tests/golden/make_assemble_goldens.py).

Arguments that only steer the reference's own solver -- ``M_inv`` (recomputed), ``NQuad``, ``is_atmos_multilayered``,
``I0_div_4pi``, ``use_banded_solver_NLayers`` -- are accepted and checked for consistency where cheap; ``autograd_compatible``
must be False (out of scope).
"""
import numpy as np

from ._engine import Plan
from .pydisort import _Closures, _tabulate_bdrf


def _bc_matrix(b, is_scalar, is_vector, N, NFourier):
    """Dirichlet boundary source in the reference's three forms -> [N, NFourier] (_solve_for_coeffs.py:142-158: a scalar
    or a vector is the zeroth Fourier mode)."""
    out = np.zeros((N, NFourier))
    if is_scalar:
        out[:, 0] = float(np.asarray(b).reshape(-1)[0]) if np.ndim(b) else float(b)
    elif is_vector:
        out[:, 0] = np.asarray(b, float).reshape(N)
    else:
        out[:, :] = np.asarray(b, float).reshape(N, NFourier)
    return out


def _assemble_intensity_and_fluxes(
    scaled_omega_arr,
    tau_arr,
    scaled_tau_arr_with_0,
    mu_arr_pos,
    M_inv, W,
    N, NQuad, NLeg,
    NFourier, NLayers, NBDRF,
    is_atmos_multilayered,
    weighted_scaled_Leg_coeffs,
    BDRF_Fourier_modes,
    mu0, I0, I0_div_4pi,
    rescale_factor, phi0,
    there_is_beam_source,
    b_pos, b_neg,
    b_pos_is_scalar, b_neg_is_scalar,
    b_pos_is_vector, b_neg_is_vector,
    Nscoeffs,
    scaled_s_poly_coeffs,
    there_is_iso_source,
    scale_tau,
    only_flux,
    use_banded_solver_NLayers,
    autograd_compatible,
    device=0,
):
    if autograd_compatible:
        raise NotImplementedError("autograd_compatible=True is outside the scope of the HIP path.")
    N, NQuad, NLeg, NFourier, NLayers, NBDRF = int(N), int(NQuad), int(NLeg), int(NFourier), int(NLayers), int(NBDRF)
    if NQuad != 2 * N or NQuad > 128:
        raise ValueError("Need NQuad = 2 N <= 128.")
    mu = np.asarray(mu_arr_pos, float).reshape(N)
    tau_arr = np.asarray(tau_arr, float).reshape(NLayers)
    beam = bool(there_is_beam_source)
    iso = bool(there_is_iso_source)
    Ns = int(Nscoeffs) if iso else 0
    # BDRF Fourier modes: scalars or callables f(mu, -mu'), tabulated on the quadrature grid as _solve_for_coeffs.py:121-134
    # evaluates them
    bq, bq0 = _tabulate_bdrf(list(BDRF_Fourier_modes)[:NBDRF], mu, mu0, beam)
    bp = _bc_matrix(b_pos, b_pos_is_scalar, b_pos_is_vector, N, NFourier)
    bn = _bc_matrix(b_neg, b_neg_is_scalar, b_neg_is_vector, N, NFourier)
    prep = dict(
        C=1, L=NLayers, N=N, P=NLeg, M=NFourier, Ns=Ns, NBDRF=NBDRF, beam=beam, mu=mu, W=np.asarray(W, float).reshape(N),
        omega_s=np.asarray(scaled_omega_arr, float).reshape(1, NLayers), tau=tau_arr[None],
        tau_s0=np.asarray(scaled_tau_arr_with_0, float).reshape(1, NLayers + 1),
        scale_tau=np.asarray(scale_tau, float).reshape(1, NLayers),
        wleg=np.asarray(weighted_scaled_Leg_coeffs, float).reshape(1, NLayers, NLeg),
        mu0=np.array([float(mu0)]), I0=np.array([float(I0)]), phi0=np.array([float(phi0)]),
        rescale=np.array([float(rescale_factor)]),
        b_pos=np.ascontiguousarray(bp.T[None]) if np.any(bp) else None,   # [C, M, N]
        b_neg=np.ascontiguousarray(bn.T[None]) if np.any(bn) else None,
        s_s=np.asarray(scaled_s_poly_coeffs, float).reshape(1, NLayers, -1)[:, :, :Ns] if Ns > 0 else None,
        bdrf_q=bq[None] if NBDRF > 0 else None, bdrf_q0=bq0[None] if NBDRF > 0 else None)
    plan = Plan(prep, device=device)
    plan.solve()
    # the closures report in the caller's units: the reference multiplies by rescale_factor (:262); the direct-beam terms of
    # `_return_act_dscale_for_reclass` (:352-374) use the beam as the caller of pydisort gave it
    sol = _Closures(plan, prep, tau_arr, NFourier, beam, float(mu0), float(I0) * float(rescale_factor))
    if only_flux:
        return sol.flux_up, sol.flux_down, sol.u0
    return sol.flux_up, sol.flux_down, sol.u0, sol.u
