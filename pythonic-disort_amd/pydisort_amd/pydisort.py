"""Drop-in for ``PythonicDISORT.pydisort`` whose hot path runs on an MI355X.

Same call signature and returned callables as the reference
(src/PythonicDISORT/pydisort.py:13-29, returns :698/:701; closure signatures
_assemble_intensity_and_fluxes.py:170, :334, :446, :527).  This module is host glue only: input
checks (pydisort.py:222-291), preparation (_prepare.py) and thin closures that call the device
evaluators.  The eigen stage, the boundary-condition solve and every evaluation of u, u0 and the
fluxes are HIP kernels reached through the C ABI in include/rtd.h; there is no CPU fallback.
"""
import threading
import warnings
from math import pi

import numpy as np

from . import _nt
from ._engine import Plan
from ._prepare import double_gauss, prepare_columns


# One-column plans whose closures are gone, kept for the next call with the same dimensions: plan creation (device arena, fills,
# stream) and the quadrature upload are a third of a one-column call's latency.  A plan is never shared: it is handed out
# when no closure of an earlier call refers to it any more (CPython frees the closures of `res = pydisort(...)` as soon as `res` is
# rebound), holds at most _IDLE_MAX plans, and a plan that was closed by hand is dropped.
# The lock is re-entrant and nothing is allocated while it is held: _release runs from _Closures.__del__, which the cyclic garbage
# collector may call on this very thread while another _release / _plan_for is inside the lock (round-5 advice).
_IDLE, _IDLE_LOCK, _IDLE_MAX = {}, threading.RLock(), 8
_IDLE_COUNT = [0]  # plans in _IDLE (kept as a counter: no generator under the lock)


def _plan_for(prep, device):
    key = (device, prep["L"], prep["N"], prep["P"], prep["M"], prep["Ns"], prep["NBDRF"], bool(prep["beam"]))
    while True:
        with _IDLE_LOCK:
            lst = _IDLE.get(key)
            plan = lst.pop() if lst else None
            if plan is not None:
                _IDLE_COUNT[0] -= 1
        if plan is None:
            break
        if getattr(plan, "_h", None):
            try:
                if getattr(plan, "_nt_on", False):
                    plan.clear_nt()
                    plan._nt_on = False
                plan.set_columns(prep)  # (same dimensions, same quadrature: only the column changes)
                return plan
            except Exception:
                plan.close()
    plan = Plan(prep, device=device)
    plan._idle_key = key
    return plan


def _release(plan):
    """Called when the closures of a call are gone: the plan waits for the next call of its shape, or is closed."""
    if not getattr(plan, "_h", None):
        return
    key = getattr(plan, "_idle_key", None)
    if key is not None:
        with _IDLE_LOCK:
            lst = _IDLE.get(key)
            if _IDLE_COUNT[0] < _IDLE_MAX and lst is not None:
                lst.append(plan)
                _IDLE_COUNT[0] += 1
                return
        if lst is None:  # first plan of this shape: the list is made outside the lock, then published under it
            fresh = [plan]
            with _IDLE_LOCK:
                if _IDLE_COUNT[0] < _IDLE_MAX and key not in _IDLE:
                    _IDLE[key] = fresh
                    _IDLE_COUNT[0] += 1
                    return
                if _IDLE_COUNT[0] < _IDLE_MAX:
                    _IDLE[key].append(plan)
                    _IDLE_COUNT[0] += 1
                    return
    plan.close()


def _tabulate_bdrf(modes, mu, mu0, beam):
    """BDRF Fourier modes (floats or callables f(mu, -mu')) -> tables q(mu_i, mu_j), q(mu_i, mu0)
    exactly as _solve_for_coeffs evaluates them (_solve_for_coeffs.py:121-134)."""
    N = len(mu)
    q = np.zeros((len(modes), N, N))
    q0 = np.zeros((len(modes), N))
    for m, f in enumerate(modes):
        if np.isscalar(f):
            q[m], q0[m] = f, f
        else:
            q[m] = f(mu, mu)
            if beam:
                q0[m] = f(mu, np.array([mu0]))[:, 0]
    return q, q0


def pydisort(
    tau_arr, omega_arr,
    NQuad,
    Leg_coeffs_all,
    mu0, I0, phi0,
    NLeg=None,
    NFourier=None,
    b_pos=0,
    b_neg=0,
    only_flux=False,
    f_arr=0,
    NT_cor=False,
    BDRF_Fourier_modes=[],
    s_poly_coeffs=np.array([[]]),
    use_banded_solver_NLayers=10,
    autograd_compatible=False,
    *,
    device=0,
):
    """Solve the 1D plane-parallel RTE for one atmospheric column; see the reference's docstring
    (pydisort.py:30-128) for the meaning of every argument.  Returns ``(mu_arr, flux_up, flux_down,
    u0[, u])``.  ``use_banded_solver_NLayers`` is validated and otherwise ignored (the device solver is
    a single block-banded elimination); ``autograd_compatible=True`` is not supported."""
    if autograd_compatible:
        raise NotImplementedError("autograd_compatible=True is not available in the MI355X build.")
    tau_arr = np.atleast_1d(np.asarray(tau_arr, dtype=float))
    omega_arr = np.atleast_1d(np.asarray(omega_arr, dtype=float))
    user_leg = Leg_coeffs_all
    Leg_coeffs_all = np.atleast_2d(np.asarray(Leg_coeffs_all, dtype=float))
    s_poly_coeffs = np.atleast_2d(np.asarray(s_poly_coeffs, dtype=float))
    f_arr = np.atleast_1d(np.asarray(f_arr, dtype=float))

    if NLeg is None:
        NLeg = NQuad
    if only_flux:
        NFourier = 1
    elif NFourier is None:
        NFourier = NQuad
    if np.all(np.asarray(b_pos) == 0):
        b_pos = 0
    if np.all(np.asarray(b_neg) == 0):
        b_neg = 0
    Nscoeffs = 0 if np.all(s_poly_coeffs == 0) else s_poly_coeffs.shape[1]
    NLayers = len(tau_arr)
    NLeg_all = Leg_coeffs_all.shape[1]
    N = NQuad // 2
    beam = I0 > 0
    iso = Nscoeffs > 0
    thickness = np.diff(tau_arr, prepend=0.0)

    # ---- input checks: same conditions and messages as pydisort.py:222-291
    if not np.all(tau_arr > 0):
        raise ValueError("tau values cannot be non-positive.")
    if not np.all(thickness > 0):
        raise ValueError("Layer thicknesses cannot be non-positive.")
    if not (np.all(omega_arr >= 0) and np.all(omega_arr < 1)):
        raise ValueError("Single-scattering albedo must be between 0 and 1, excluding 1.")
    if not NLeg > 0:
        raise ValueError("The number of phase function Legendre coefficients must be positive.")
    if not NLeg <= NLeg_all:
        raise ValueError("`NLeg` cannot be larger than the number of phase function Legendre coefficients provided.")
    if not Leg_coeffs_all.shape[0] == NLayers:
        raise ValueError("The zeroth dimension of the shape of `Leg_coeffs_all` does not match the number of layers which is deduced from the length of `tau_arr`.")
    if not len(omega_arr) == NLayers:
        raise ValueError("The zeroth dimension of the shape of `omega_arr` does not match the number of layers which is deduced from the length of `tau_arr`.")
    if np.any(f_arr != 0) and not len(f_arr) == NLayers:
        raise ValueError("The length of `f_arr` does not match the number of layers which is deduced from the length of `tau_arr`.")
    if iso and not s_poly_coeffs.shape[0] == NLayers:
        raise ValueError("The zeroth dimension of the shape of `s_poly_coeffs` does not match the number of layers which is deduced from the length of `tau_arr`.")
    if not np.all(omega_arr * Leg_coeffs_all[:, 0] == omega_arr):
        warnings.warn("The zeroth index phase function Legendre coefficient must be, and has been corrected to, 1.")
        Leg_coeffs_all[:, 0] = 1
        if isinstance(user_leg, np.ndarray) and user_leg.dtype == float:  # the reference fixes the caller's array in place
            np.atleast_2d(user_leg)[:, 0] = 1
    if not (np.all(-1 < Leg_coeffs_all[:, 1:]) and np.all(Leg_coeffs_all[:, 1:] < 1)):
        raise ValueError("The phase function Legendre coefficients must all be between -1 and 1 exclusive (only the zeroth coefficient can equal 1).")
    if not NQuad >= 2:
        raise ValueError("There must be at least two streams.")
    if not NQuad % 2 == 0:
        raise ValueError("The number of streams must be even.")
    if not NFourier > 0:
        raise ValueError("The number of Fourier modes to use in the solution must be positive.")
    if not NFourier <= NLeg:
        raise ValueError("The number of Fourier modes to use in the solution must be less than or equal to the number of phase function Legendre coefficients used.")
    if NFourier > 64 and not only_flux:
        warnings.warn("`NFourier` is large and may cause errors, consider decreasing `NFourier` to 64 and it probably should be even less. By default `NFourier` equals `NQuad`.")
    if not NLeg <= NQuad:
        raise ValueError("There should be more streams than the number of phase function Legendre coefficients used.")
    if I0 < 0:
        raise ValueError("The intensity of the incident beam cannot be negative.")
    if beam:
        if not (0 < mu0 and mu0 <= 1):
            raise ValueError("The cosine of the polar angle of the incident beam must be between 0 and 1, excluding 0.")
        if not (0 <= phi0 and phi0 < 2 * pi):
            raise ValueError("Provide the principal azimuthal angle for the incident beam (must be between 0 and 2pi, excluding 2pi).")

    def bc_matrix(b, what):  # scalar | [N] | [N, NFourier]  ->  [N, NFourier]  (_solve_for_coeffs.py:142-158)
        out = np.zeros((N, NFourier))
        b = np.asarray(b, dtype=float)
        if len(np.atleast_1d(b)) == 1:
            out[:, 0] = float(np.atleast_1d(b).reshape(-1)[0])
        elif len(b) == N and b.ndim == 1:
            out[:, 0] = b
        elif b.shape == (N, NFourier):
            out[:] = b
        else:
            raise ValueError(f"The shape of the {what} boundary condition is incorrect.")
        return out

    b_pos_m = bc_matrix(b_pos, "bottom")
    b_neg_m = bc_matrix(b_neg, "top")
    if not (np.all(0 <= f_arr) and np.all(f_arr <= 1)):
        raise ValueError("The fractional scattering must be between 0 and 1.")
    if not use_banded_solver_NLayers >= 3:
        raise ValueError("The minimum threshold `use_banded_solver_NLayers` is 3, else the matrix will not be banded.")
    if NQuad > 128:  # (66 ... 128 streams: the NP = 64 kernels, csrc/rtd_bc_wide.hip and rtd_eigen_kernel<64, 2>)
        raise ValueError("This build supports at most 128 streams (NQuad <= 128).")

    mu_pos, W = double_gauss(N)
    mu_arr = np.concatenate([mu_pos, -mu_pos])
    if NT_cor and np.any(np.abs(mu_pos - mu0) < 1e-8):
        raise ValueError("Some quadrature angles come too close to `mu0`. Perturb `NQuad` or `mu0` to rectify this error.")

    bq, bq0 = _tabulate_bdrf(BDRF_Fourier_modes, mu_pos, mu0, beam)
    f_full = np.broadcast_to(f_arr, (NLayers,)) if len(f_arr) == 1 else f_arr
    prep = prepare_columns(
        tau_arr[None], omega_arr[None], NQuad, Leg_coeffs_all[None], [mu0], [I0], [phi0], NLeg, NFourier,
        b_pos_m[None], b_neg_m[None], f_full[None],
        s_poly_coeffs[None, :, :Nscoeffs] if iso else np.zeros((1, NLayers, 0)), bq[None], bq0[None])
    if np.any(prep["omega_s"] > 1 - 1e-6):
        warnings.warn("Some delta-scaled single-scattering albedos are very close to 1 which may cause numerical instability.")
    if np.any(-0.95 > prep["leg_s"][0, :, 1:]) or np.any(prep["leg_s"][0, :, 1:] > 0.95):
        warnings.warn("Some delta-scaled phase function Legendre coefficients have a magnitude that is very close to 1"
                      " (this excludes the zeroth index coefficient which must be 1) which may cause numerical instability.")

    plan = _plan_for(prep, device)
    try:
        plan.solve()
    except BaseException:  # a plan whose solve could not even be queued is not handed to the next call: closed here, not left to __del__
        plan.close()
        raise
    sol = _Closures(plan, prep, tau_arr, NFourier, beam, mu0, I0)

    if only_flux:
        return mu_arr, sol.flux_up, sol.flux_down, sol.u0
    nt_on = (NT_cor and beam and np.any(f_arr > 0) and NLeg < NLeg_all and np.any(omega_arr > 0))
    if nt_on:  # the device adds TMS + IMS to u from now on (the reference returns u_corrected as `u`, :696-698)
        plan.set_nt(*_nt.nt_inputs(prep, omega_arr[None], f_full[None], Leg_coeffs_all[None], NLeg, [mu0]))
        plan._nt_on = True
    return mu_arr, sol.flux_up, sol.flux_down, sol.u0, sol.u


class _Closures:
    """The returned callables.  They keep the device plan (GC, K, B ... stay in HBM) alive."""

    def __init__(self, plan, prep, tau_arr, M, beam, mu0, I0_user):
        self.plan, self.prep, self.tau_arr, self.M, self.beam = plan, prep, tau_arr, M, beam
        self.mu0, self.I0_user = mu0, I0_user

    def __del__(self):  # the last of the returned callables is gone: the plan may serve the next call of this shape
        try:
            _release(self.plan)
        except Exception:
            pass

    def _tau(self, tau):
        tau = np.atleast_1d(np.asarray(tau, dtype=float))
        if np.any(tau < 0) or np.any(tau > self.tau_arr[-1]):
            raise ValueError("tau input outside the tau range specified for the atmosphere (check `tau_arr`).")
        return tau

    def u(self, tau, phi, is_antiderivative_wrt_tau=False, return_Fourier_error=False, return_tau_arr=False,
          *, _return_l=False):
        tau = self._tau(tau)
        phi = np.atleast_1d(np.asarray(phi, dtype=float))
        r = self.plan.evaluate(tau[None], phi, is_antiderivative_wrt_tau, want=("u",))
        outs = (np.squeeze(r["u"][0]),)
        if return_Fourier_error:  # _assemble.py:264-318; measured on the uncorrected delta-M solution (pydisort.py:652-660)
            r = self.plan.evaluate(tau[None], phi, is_antiderivative_wrt_tau, want=("u", "ulast"), skip_nt=True)
            ua = np.abs(r["u"][0])
            last = np.abs(r["ulast"][0][:, :, None] * np.cos((self.M - 1) * (self.prep["phi0"][0] - phi))[None, None, :])
            outs += (np.max(np.divide(last, ua, out=np.zeros_like(ua), where=ua > 1e-8 * self.prep["rescale"][0])),)
        if return_tau_arr:
            outs += (self.tau_arr,)
        if _return_l:
            outs += (np.argmax(tau[:, None] <= self.tau_arr[None, :], axis=1),)
        return outs if len(outs) > 1 else outs[0]

    def u0(self, tau, is_antiderivative_wrt_tau=False, return_tau_arr=False, _return_act_dscale_for_reclass=False):
        tau = self._tau(tau)
        r = self.plan.evaluate(tau[None], None, is_antiderivative_wrt_tau, want=("u0",))
        outs = (np.squeeze(r["u0"][0]),)
        if return_tau_arr:
            outs += (self.tau_arr,)
        if _return_act_dscale_for_reclass:  # _assemble.py:352-374
            p = self.prep
            if np.any(p["scale_tau"][0] != 1):
                l = np.argmax(tau[:, None] <= self.tau_arr[None, :], axis=1)
                ts = p["tau_s0"][0, 1:][l] - (self.tau_arr[l] - tau) * p["scale_tau"][0, l]
                if is_antiderivative_wrt_tau:
                    rec = (self.I0_user * np.exp(-ts / self.mu0) / (-p["scale_tau"][0, l] / self.mu0)
                           - self.I0_user * np.exp(-tau / self.mu0) * -self.mu0)
                else:
                    rec = self.I0_user * np.exp(-ts / self.mu0) - self.I0_user * np.exp(-tau / self.mu0)
            else:
                rec = 0
            outs += (rec,)
        return outs if len(outs) > 1 else outs[0]

    def flux_up(self, tau, is_antiderivative_wrt_tau=False, return_tau_arr=False):
        tau = self._tau(tau)
        r = self.plan.evaluate(tau[None], None, is_antiderivative_wrt_tau, want=("flux",))
        out = np.squeeze(r["flux_up"][0])[()]
        return (out, self.tau_arr) if return_tau_arr else out

    def flux_down(self, tau, is_antiderivative_wrt_tau=False, return_tau_arr=False):
        tau = self._tau(tau)
        r = self.plan.evaluate(tau[None], None, is_antiderivative_wrt_tau, want=("flux",))
        # without a beam source the reference's direct flux is the scalar 0 (_assemble.py:568-570, :610)
        direct = np.squeeze(r["flux_down_direct"][0])[()] if self.beam else np.float64(0.0)
        outs = (np.squeeze(r["flux_down_diffuse"][0])[()], direct)
        return outs + (self.tau_arr,) if return_tau_arr else outs
