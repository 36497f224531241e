"""Column-batched entry point: many independent atmospheres per call (no reference equivalent -- the
reference solves one column per ``pydisort`` call)."""
import numpy as np

from ._engine import Plan
from . import _nt
from ._prepare import double_gauss, prepare_columns


DEFAULT_RETAIN_BYTES = 16 << 30  # pydisort_batch(retain=..., retain_bytes=None): what a call may keep on the device for its evaluators


class BatchSolution:
    """Evaluators over all columns; arrays carry a leading column axis."""

    def __init__(self, plan, prep):
        self.plan, self.prep = plan, prep
        self.mu_arr = np.concatenate((prep["mu"], -prep["mu"]))

    def _tau(self, tau):
        tau = np.asarray(tau, dtype=float)
        if tau.ndim == 1:
            tau = np.broadcast_to(tau, (self.prep["C"], len(tau)))
        if np.any(tau < 0) or np.any(tau > self.prep["tau"][:, -1:]):
            raise ValueError("tau input outside the tau range specified for the atmosphere (check `tau_arr`).")
        return np.ascontiguousarray(tau)

    def u(self, tau, phi, is_antiderivative_wrt_tau=False):
        """-> [C, NQuad, ntau, nphi]"""
        return self.plan.evaluate(self._tau(tau), phi, is_antiderivative_wrt_tau, want=("u",))["u"]

    def u0(self, tau, is_antiderivative_wrt_tau=False):
        """-> [C, NQuad, ntau]"""
        return self.plan.evaluate(self._tau(tau), None, is_antiderivative_wrt_tau, want=("u0",))["u0"]

    def flux_up(self, tau, is_antiderivative_wrt_tau=False):
        """-> [C, ntau]"""
        return self.plan.evaluate(self._tau(tau), None, is_antiderivative_wrt_tau, want=("flux",))["flux_up"]

    def flux_down(self, tau, is_antiderivative_wrt_tau=False):
        """-> (diffuse [C, ntau], direct [C, ntau])"""
        r = self.plan.evaluate(self._tau(tau), None, is_antiderivative_wrt_tau, want=("flux",))
        return r["flux_down_diffuse"], r["flux_down_direct"]


def pydisort_batch(tau_arr, omega_arr, NQuad, Leg_coeffs_all, mu0, I0, phi0, NLeg=None, NFourier=None,
                   b_pos=0, b_neg=0, only_flux=False, f_arr=0, NT_cor=False, bdrf_q=None, bdrf_q0=None,
                   s_poly_coeffs=None, device=0, bdrf_samples=None, NBDRF=None, mode_shard=None, work_columns=0,
                   device_prepare=False, numeric_errors="raise", retain="auto", retain_bytes=None, _defer_solve=False):
    """Like ``pydisort`` with a leading column axis on every atmospheric input:
    tau_arr, omega_arr, f_arr [C, L]; Leg_coeffs_all [C, L, NLeg_all]; mu0, I0, phi0 [C];
    b_pos / b_neg: scalar, [C], [C, N] or [C, N, NFourier]; s_poly_coeffs [C, L, Ns];
    bdrf_q [C, NBDRF, N, N] and bdrf_q0 [C, NBDRF, N]: BDRF Fourier modes tabulated on the quadrature grid.
    bdrf_samples=(rho_qq [C, N, N, nphi], rho_q0 [C, N, nphi] or None) instead: the reflectance itself sampled at
    dphi_p = 2 pi p / nphi; its first NBDRF (default NFourier) Fourier modes are then formed on the device
    (``subroutines.sample_BDRF`` builds the samples from a function rho(mu, mu', dphi)).
    NT_cor=True adds the Nakajima-Tanaka corrections to ``u`` on the device (needs a beam in every column,
    f_arr > 0 and more Legendre coefficients than NLeg).
    mode_shard=(r, G): solve only the Fourier modes r, r + G, r + 2G, ... (the partition for
    fewer columns than GPUs); the evaluators then return this shard's partial sums -- the shards add up to the full
    result (u0, fluxes and NT corrections come from shard 0 only; ``Plan.allreduce_results`` sums across RCCL ranks).
    device_prepare=True: the delta-M scaling and the source rescaling of pydisort.py:316-372 run on the device from the raw
    inputs (``rtd_plan_set_columns_raw``) instead of in NumPy -- for throughput batches; not with NT_cor or mode_shard.
    numeric_errors: "raise" (a column whose solve fails numerically -- e.g. a phase function whose truncation is not
    positive -- raises NumericalError from the evaluators; ``sol.plan.column_status()`` tells which) or "nan" (the failed
    columns are NaN in the returned arrays, every other column keeps its result: one bad column does not cost the batch).
    work_columns: columns whose intermediates are resident on the device at a time (0: sized by the library); batches
    larger than that are solved window by window (include/rtd.h: rtd_plan_create_windowed).
    retain: the returned evaluators keep what they need of the solve for EVERY column -- as the reference's closures keep
    GC_collect, K_collect, B_collect (_assemble_intensity_and_fluxes.py:170-262) -- so that calling them again costs an
    evaluation, not a solve, also for a batch of several windows (include/rtd.h: rtd_plan_create_retained).  Two forms: "full"
    (3.1 MB per 20-layer 32-stream column: an evaluator call only evaluates) and "lean" (0.5 MB per such column: coefficients,
    eigenvalues and particular solutions stay, an evaluator call re-runs the eigen stage -- not the boundary-condition solve -- for
    the layers its points touch; same bits, about a tenth of a solve for one depth per column; 10 streams and up).  "auto"
    (default): full while it fits the budget, else lean while that fits, else neither; "full" / "lean": that form or neither;
    False: never (every call of an evaluator on a batch of several windows then solves them again).
    retain_bytes: the budget in bytes.  None (default): DEFAULT_RETAIN_BYTES = 16 GiB, and never more than three tenths of the
    free device memory -- the library is a guest on the GPU; a caller that wants BASELINE's 10^5-column batch retained passes what
    it is willing to give (50 GB lean).
    All columns share NQuad, NLeg, NFourier and the layer count.  Returns (mu_arr, BatchSolution)."""
    tau_arr = np.atleast_2d(np.asarray(tau_arr, float))
    C, L = tau_arr.shape
    omega_arr = np.broadcast_to(np.asarray(omega_arr, float), (C, L))
    Leg = np.asarray(Leg_coeffs_all, float)
    if Leg.ndim == 2:
        Leg = np.broadcast_to(Leg[None], (C,) + Leg.shape)
    N = NQuad // 2
    NLeg = NQuad if NLeg is None else NLeg
    NFourier = 1 if only_flux else (NQuad if NFourier is None else NFourier)
    if NQuad % 2 or NQuad < 2 or NQuad > 128:
        raise ValueError("NQuad must be even and between 2 and 128.")
    if not (0 < NFourier <= NLeg <= NQuad and NLeg <= Leg.shape[2]):
        raise ValueError("Need 0 < NFourier <= NLeg <= NQuad and NLeg <= number of Legendre coefficients provided.")
    if not (np.all(tau_arr > 0) and np.all(np.diff(tau_arr, axis=1) > 0)):
        raise ValueError("tau values must be positive and increasing.")
    if not (np.all(omega_arr >= 0) and np.all(omega_arr < 1)):
        raise ValueError("Single-scattering albedo must be between 0 and 1, excluding 1.")
    mu0 = np.broadcast_to(np.asarray(mu0, float), (C,))
    I0 = np.broadcast_to(np.asarray(I0, float), (C,))
    phi0 = np.broadcast_to(np.asarray(phi0, float), (C,))
    if np.any(I0 < 0) or (np.any(I0 > 0) and not np.all((mu0 > 0) & (mu0 <= 1))):
        raise ValueError("Need I0 >= 0 and 0 < mu0 <= 1 for every column when there is a beam source.")

    def bc(b):
        b = np.asarray(b, float)
        if b.ndim == 0 and b == 0:
            return None  # all zero: nothing to allocate or upload
        out = np.zeros((C, N, NFourier))
        if b.ndim == 0 or b.shape == (C,):
            out[:, :, 0] = np.broadcast_to(b, (C,))[:, None]
        elif b.shape == (C, N):
            out[:, :, 0] = b
        elif b.shape == (C, N, NFourier):
            out[:] = b
        else:
            raise ValueError("The shape of a boundary condition is incorrect.")
        return out

    f_arr = np.broadcast_to(np.asarray(f_arr, float), (C, L))
    sp = np.zeros((C, L, 0)) if s_poly_coeffs is None or np.all(np.asarray(s_poly_coeffs) == 0) \
        else np.asarray(s_poly_coeffs, float).reshape(C, L, -1)
    bq = np.zeros((C, 0, N, N)) if bdrf_q is None else np.asarray(bdrf_q, float)
    bq0 = np.zeros((C, 0, N)) if bdrf_q0 is None else np.asarray(bdrf_q0, float)
    if bdrf_samples is not None:
        if bdrf_q is not None:
            raise ValueError("Give either bdrf_q / bdrf_q0 or bdrf_samples, not both.")
        nb = NFourier if NBDRF is None else int(NBDRF)
        if not 0 < nb <= NFourier:
            raise ValueError("Need 0 < NBDRF <= NFourier.")
        bq, bq0 = np.zeros((C, nb, N, N)), np.zeros((C, nb, N))  # placeholders: the device fills the tables
    if device_prepare:
        if NT_cor or mode_shard is not None:
            raise ValueError("device_prepare cannot be combined with NT_cor or mode_shard.")
        bp, bn = bc(b_pos), bc(b_neg)
        raw = dict(tau_arr=tau_arr, omega_arr=omega_arr, leg=Leg, f_arr=f_arr, mu0=mu0, I0=I0, phi0=phi0,
                   b_pos=None if bp is None else np.ascontiguousarray(bp.transpose(0, 2, 1)),
                   b_neg=None if bn is None else np.ascontiguousarray(bn.transpose(0, 2, 1)),
                   s_poly=sp if sp.shape[2] > 0 else None,
                   bdrf_q=bq if bq.shape[1] > 0 else None, bdrf_q0=bq0 if bq.shape[1] > 0 else None)
        mu, W = double_gauss(N)
        prep = dict(C=C, L=L, N=N, P=NLeg, M=NFourier, Ns=sp.shape[2], NBDRF=bq.shape[1], beam=bool(np.any(I0 > 0)),
                    mu=mu, W=W, tau=tau_arr, raw=raw)
    else:
        prep = prepare_columns(tau_arr, omega_arr, NQuad, Leg, mu0, I0, phi0, NLeg, NFourier, bc(b_pos), bc(b_neg),
                               f_arr, sp, bq, bq0)
    if mode_shard is not None:
        r, G = int(mode_shard[0]), int(mode_shard[1])
        if not (0 <= r < G <= NFourier):
            raise ValueError("mode_shard=(r, G) needs 0 <= r < G <= NFourier.")
        modes = np.arange(r, NFourier, G)
        for k in ("b_pos", "b_neg"):  # the source rescale above saw every mode
            if prep[k] is not None:
                prep[k] = np.ascontiguousarray(prep[k][:, modes, :])
        prep["M"] = len(modes)
        prep["mode_shard"] = (r, G, NFourier)
        NT_cor = NT_cor and r == 0
    if numeric_errors not in ("raise", "nan"):
        raise ValueError('numeric_errors must be "raise" or "nan".')
    form = {"auto": 0, True: 0, "full": 1, "lean": 2}.get(retain if isinstance(retain, (str, bool)) else "auto")
    if retain is False or retain is None or (retain == "auto" and _defer_solve):
        budget = 0  # (the throughput callers drive plan.run() themselves: nothing to keep)
    elif form is None:
        raise ValueError('retain must be "auto", "full", "lean", False or a number of bytes.')
    elif isinstance(retain, (int, np.integer)) and not isinstance(retain, bool):
        budget = int(retain)  # (an int: that many bytes, the round-5 spelling of retain_bytes)
    else:
        budget = DEFAULT_RETAIN_BYTES if retain_bytes is None else int(retain_bytes)
        if budget > 0 and retain_bytes is None:
            # the default never takes more than three tenths of what is free -- asked of the runtime only when the batch is large
            # enough for it to matter (hipMemGetInfo costs a small call a noticeable fraction of its time)
            NP = 4
            while NP < N:
                NP *= 2
            if C * NFourier * L * (2 * NP * NP + 8 * NP) * 8 > (256 << 20):
                budget = max(1, min(budget, int(0.3 * Plan.free_device_bytes(device))))
    plan = Plan(prep, device=device, work_columns=work_columns, retain_bytes=budget, retain_form=form or 0)
    plan.numeric_errors = numeric_errors
    if bdrf_samples is not None:
        plan.set_bdrf_samples(bdrf_samples[0], bdrf_samples[1] if np.any(I0 > 0) else None)
    if not _defer_solve:
        plan.solve()
    if NT_cor and not only_flux:
        if not (np.all(I0 > 0) and np.any(f_arr > 0) and NLeg < Leg.shape[2]):
            raise ValueError("NT_cor needs a beam source in every column, f_arr > 0 and NLeg < number of Legendre coefficients.")
        if np.any(np.abs(prep["mu"][None, :] - mu0[:, None]) < 1e-8):
            raise ValueError("Some quadrature angles come too close to `mu0`. Perturb `NQuad` or `mu0` to rectify this error.")
        plan.set_nt(*_nt.nt_inputs(prep, omega_arr, f_arr, Leg, NLeg, mu0))
    sol = BatchSolution(plan, prep)
    return sol.mu_arr, sol


def solve_columns_streamed(cfg, tau, phi, chunk_columns=0, device=0, only_flux=False, out=None, numeric_errors="raise"):
    """Throughput form for large column counts: ONE plan holds the inputs and the results of all columns, the
    intermediates of the solve (~8 MB per cfg4 column) live for `chunk_columns` columns at a time (0: ~8 192 (column,
    mode) chains per window) and the device-to-host copies of a window overlap the kernels of the next (``Plan.run_fetch``).
    Source terms are taken from the whole batch (a beam or a thermal source in any column switches it on for all).

    cfg : dict of ``pydisort_batch`` keyword arguments with a leading column axis (tau_arr, omega_arr, Leg_coeffs_all,
          mu0, I0, phi0, optional f_arr, b_pos, b_neg, s_poly_coeffs, bdrf_q, bdrf_q0, NLeg, NFourier); NQuad scalar.
    tau : [C, ntau] evaluation depths; phi : [nphi].
    out : optional dict of preallocated C-contiguous float64 result arrays to fill (the same keys and shapes).
    numeric_errors : as in ``pydisort_batch`` ("nan": failed columns come back as NaN, the rest of the batch is kept).
    Returns dict(u [C, NQuad, ntau, nphi] (absent when only_flux), u0, flux_up, flux_down_diffuse, flux_down_direct)."""
    tau = np.ascontiguousarray(np.asarray(tau, float))
    C, ntau = tau.shape
    if chunk_columns <= 0:
        # windows of ~8 192 (column, mode) chains: the two-stream window pipeline of the library does best there (256 cfg4
        # columns: +4 % over windows of 2 048); the library shrinks a window that does not fit the
        # device (RTD_WORK_BYTES when set, else 80 % of the free device memory: rtd_api.hip plan_build)
        nq = int(cfg["NQuad"])
        modes = 1 if only_flux else int(cfg.get("NFourier") or cfg.get("NLeg") or nq)
        chunk_columns = max(64, 8192 // max(modes, 1))
    _, sol = pydisort_batch(only_flux=only_flux, device=device, work_columns=chunk_columns, device_prepare=True,
                            numeric_errors=numeric_errors, _defer_solve=True, **cfg)
    plan = sol.plan
    try:
        sol._tau(tau)  # range check on the host, with the reference's message
        phi = np.array([0.0]) if only_flux else np.atleast_1d(np.asarray(phi, float))
        plan.set_eval_points(tau, phi)
        Q = plan.Q
        shapes = dict(u0=(C, Q, ntau), flux_up=(C, ntau), flux_down_diffuse=(C, ntau), flux_down_direct=(C, ntau))
        if not only_flux:
            shapes["u"] = (C, Q, ntau, len(phi))
        if out is None:
            out = {k: np.empty(shp) for k, shp in shapes.items()}
        else:
            for k, shp in shapes.items():
                a = out.get(k)
                if a is None or a.shape != shp or a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"]:
                    raise ValueError(f"out[{k!r}] must be a C-contiguous float64 array of shape {shp}.")
        plan.run_fetch(out)
    finally:
        plan.close()
    return out
