"""Thin object wrapper over a device plan (include/rtd.h): uploads prepared columns, launches the HIP
path, evaluates at (tau, phi).  Host arrays are NumPy float64; nothing here computes on the CPU."""
import ctypes as C

import numpy as np

from . import _lib


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class Plan:
    def __init__(self, prep, device=0, work_columns=0, retain_bytes=0, retain_form=0):
        """prep: dict from _prepare.prepare_columns.  work_columns: columns whose intermediates are resident at a time
        (0: sized by the library, include/rtd.h: rtd_plan_create_windowed).  retain_bytes: budget for keeping what the
        evaluators need of a solve for ALL columns of a plan of several windows, so that evaluate() does not solve again
        (include/rtd.h: rtd_plan_create_retained; -1: three tenths of the free device memory, 0: never -- the throughput form).
        retain_form: 0 the full state, else the lean one (coefficients, k, E, B; Y, A recomputed per evaluation), 1 full only,
        2 lean (rtd_plan_create_retained_form)."""
        lib = _lib.load()
        self._lib = lib
        self.prep = prep
        self.C, self.L, self.N, self.Q = prep["C"], prep["L"], prep["N"], 2 * prep["N"]
        self.M = prep["M"]
        dims = _lib.rtd_dims(prep["C"], prep["L"], 2 * prep["N"], prep["P"], prep["M"], prep["Ns"],
                             prep["NBDRF"], int(prep["beam"]))
        h = C.c_void_p()
        _lib.check(lib.rtd_plan_create_retained_form(C.byref(dims), device, int(work_columns), int(retain_bytes), int(retain_form), C.byref(h)))
        self._h = h
        mu, w = _f64(prep["mu"]), _f64(prep["W"])
        _lib.check(lib.rtd_plan_set_quadrature(h, _lib.dptr(mu), _lib.dptr(w)))
        if prep.get("raw") is not None:
            self.set_columns_raw(prep["raw"])
        else:
            self.set_columns(prep)
        if prep.get("mode_shard") is not None:
            self.set_mode_shard(*prep["mode_shard"])

    def set_mode_shard(self, first, stride, total):
        """The plan's M local Fourier modes stand for the modes first, first + stride, ... of `total`
        (include/rtd.h: rtd_plan_set_mode_shard); evaluators then return this shard's partial sums."""
        _lib.check(self._lib.rtd_plan_set_mode_shard(self._h, int(first), int(stride), int(total)))
        self.solved = False

    def allreduce_results(self):
        """RCCL all-reduce (sum) of the results of run() over the ranks of the communicator (mode shards)."""
        _lib.check(self._lib.rtd_comm_allreduce_results(self._h))

    def set_columns(self, prep):
        """Upload the prepared per-column inputs (same dimensions as the plan): a plan can be reused for many batches.
        A plan created without a beam source skips the beam terms on the device, so a batch that has one is refused
        (and a batch without one for a plan with one is fine: its beam terms are zero)."""
        for k in ("C", "L", "N", "P", "M", "Ns", "NBDRF"):
            if prep[k] != self.prep[k]:
                raise ValueError(f"prepared batch does not match the plan: {k} = {prep[k]} vs {self.prep[k]}")
        if prep["beam"] and not self.prep["beam"]:
            raise ValueError("prepared batch has a beam source but the plan was created without one")
        keys = ["omega_s", "tau", "tau_s0", "scale_tau", "wleg", "mu0", "I0", "phi0", "rescale",
                "b_pos", "b_neg", "s_s", "bdrf_q", "bdrf_q0"]
        arrs = [_f64(prep[k]) for k in keys]
        _lib.check(self._lib.rtd_plan_set_columns(self._h, *[_lib.dptr(a) for a in arrs]))
        self.prep = prep
        self.solved = False

    def set_columns_raw(self, raw):
        """Upload a batch as the user gave it; delta-M scaling and source rescaling run on the device
        (include/rtd.h: rtd_plan_set_columns_raw).  raw: dict(tau_arr, omega_arr, f_arr [C, L], leg [C, L, nleg_all],
        mu0, I0, phi0 [C], b_pos / b_neg [C, M, N] or None, s_poly [C, L, Ns] or None, bdrf_q [C, NBDRF, N, N],
        bdrf_q0 [C, NBDRF, N] or None).  The C ABI takes bare pointers whose extents the plan implies, so every
        array's shape is checked here."""
        names = ("tau_arr", "omega_arr", "leg", "f_arr", "mu0", "I0", "phi0", "b_pos", "b_neg", "s_poly", "bdrf_q", "bdrf_q0")
        a = {k: _f64(raw.get(k)) for k in names}
        p = self.prep
        Cn, L, N, M, Ns, NB = p["C"], p["L"], p["N"], p["M"], p["Ns"], p["NBDRF"]
        want = {"tau_arr": (Cn, L), "omega_arr": (Cn, L), "f_arr": (Cn, L), "mu0": (Cn,), "I0": (Cn,), "phi0": (Cn,),
                "b_pos": (Cn, M, N), "b_neg": (Cn, M, N), "s_poly": (Cn, L, Ns), "bdrf_q": (Cn, NB, N, N),
                "bdrf_q0": (Cn, NB, N)}
        required = {"tau_arr", "omega_arr", "f_arr", "mu0", "I0", "phi0"} | ({"s_poly"} if Ns > 0 else set()) \
            | ({"bdrf_q", "bdrf_q0"} if NB > 0 else set())
        if a["leg"] is None or a["leg"].ndim != 3 or a["leg"].shape[:2] != (Cn, L) or a["leg"].shape[2] < p["P"]:
            raise ValueError(f"raw batch does not match the plan: leg must be [C = {Cn}, L = {L}, nleg_all >= {p['P']}]")
        for k, shape in want.items():
            if a[k] is None:
                if k in required:
                    raise ValueError(f"raw batch does not match the plan: {k} is required")
                continue
            if (k == "s_poly" and Ns == 0) or (k.startswith("bdrf") and NB == 0):
                a[k] = None  # the plan has no such term; the library ignores the pointer
                continue
            if a[k].shape != shape:
                raise ValueError(f"raw batch does not match the plan: {k} has the shape {a[k].shape}, expected {shape}")
        if not p["beam"] and np.any(a["I0"] > 0):
            raise ValueError("raw batch has a beam source but the plan was created without one")
        _lib.check(self._lib.rtd_plan_set_columns_raw(
            self._h, _lib.dptr(a["tau_arr"]), _lib.dptr(a["omega_arr"]), _lib.dptr(a["leg"]), a["leg"].shape[2],
            *[_lib.dptr(a[k]) for k in ("f_arr", "mu0", "I0", "phi0", "b_pos", "b_neg", "s_poly", "bdrf_q", "bdrf_q0")]))
        # what later checks (set_columns, the closures' tau range, beam terms) compare against follows the new batch
        self.prep = dict(p, tau=a["tau_arr"], mu0=a["mu0"], phi0=a["phi0"], raw=raw)
        self.solved = False

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rtd_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bdrf_samples(self, rho_qq, rho_q0=None):
        """Form the plan's NBDRF BDRF Fourier modes on the device from samples of the reflectance on a uniform grid
        of relative azimuths dphi_p = 2 pi p / nphi:  rho_qq [C, N, N, nphi] = rho(mu_i, mu_j, dphi_p) and, with a
        beam, rho_q0 [C, N, nphi] = rho(mu_i, mu0, dphi_p).  Replaces the tables given to ``set_columns``."""
        Cn, N = self.prep["C"], self.prep["N"]
        rho_qq = _f64(rho_qq)
        if rho_qq.ndim != 4 or rho_qq.shape[:3] != (Cn, N, N):
            raise ValueError("rho_qq must have the shape [C, N, N, nphi].")
        nphi = rho_qq.shape[3]
        q0 = None
        if rho_q0 is not None:
            q0 = _f64(rho_q0)
            if q0.shape != (Cn, N, nphi):
                raise ValueError("rho_q0 must have the shape [C, N, nphi].")
        _lib.check(self._lib.rtd_plan_set_bdrf_samples(self._h, nphi, _lib.dptr(rho_qq), _lib.dptr(q0) if q0 is not None else None))
        self.solved = False

    def set_nt(self, weighted_leg_all, f_arr, ims_coef, ims_par):
        """Enable device-side Nakajima-Tanaka corrections of `u` (arrays with a leading column axis)."""
        a = [_f64(v) for v in (weighted_leg_all, f_arr, ims_coef, ims_par)]
        _lib.check(self._lib.rtd_plan_set_nt(self._h, a[0].shape[-1], *[_lib.dptr(v) for v in a]))

    def clear_nt(self):
        _lib.check(self._lib.rtd_plan_set_nt(self._h, 0, None, None, None, None))

    def solve(self):
        _lib.check(self._lib.rtd_plan_solve(self._h))
        self.solved = True

    def synchronize(self):
        _lib.check(self._lib.rtd_plan_synchronize(self._h))

    def device_bytes(self):
        b = C.c_int64()
        _lib.check(self._lib.rtd_plan_device_bytes(self._h, C.byref(b)))
        return b.value

    @staticmethod
    def pool_bytes():
        """Device memory of closed plans that the library keeps for the next plan (include/rtd.h: rtd_pool_bytes)."""
        b = C.c_int64()
        _lib.check(_lib.load().rtd_pool_bytes(C.byref(b)))
        return b.value

    @staticmethod
    def pool_trim(device=-1):
        """Gives that memory back to the runtime (include/rtd.h: rtd_pool_trim); returns the bytes released."""
        b = C.c_int64()
        _lib.check(_lib.load().rtd_pool_trim(int(device), C.byref(b)))
        return b.value

    @staticmethod
    def free_device_bytes(device=0):
        """Free memory of the device right now (include/rtd.h: rtd_device_memory)."""
        b = C.c_int64()
        _lib.check(_lib.load().rtd_device_memory(int(device), C.byref(b), None))
        return b.value

    @staticmethod
    def pool_set_limit(nbytes=-1, device=0):
        """Opt in to (or out of) keeping the LARGE device blocks of closed plans for the next plan: nbytes per device, < 0 = an
        eighth of the device's memory, 0 = off (the default).  Returns the previous limit (include/rtd.h: rtd_pool_set_limit)."""
        b = C.c_int64()
        _lib.check(_lib.load().rtd_pool_set_limit(int(nbytes), int(device), C.byref(b)))
        return b.value

    def evaluate(self, tau, phi=None, antiderivative=False, want=("u", "u0", "flux"), skip_nt=False):
        """tau [C, ntau]; phi [nphi] or None -> dict of arrays (u [C,Q,ntau,nphi], u0 [C,Q,ntau],
        flux_up / flux_down_diffuse / flux_down_direct [C,ntau], ulast [C,Q,ntau])."""
        tau = _f64(np.atleast_2d(tau))
        assert tau.shape[0] == self.C
        ntau = tau.shape[1]
        phi = _f64(np.atleast_1d(phi)) if phi is not None else None
        nphi = 0 if phi is None else len(phi)
        out = {}
        u = np.empty((self.C, self.Q, ntau, nphi)) if ("u" in want and nphi > 0) else None
        u0 = np.empty((self.C, self.Q, ntau)) if "u0" in want else None
        ul = np.empty((self.C, self.Q, ntau)) if "ulast" in want else None
        fl = [np.empty((self.C, ntau)) for _ in range(3)] if "flux" in want else [None] * 3
        out.update(u=u, u0=u0, ulast=ul, flux_up=fl[0], flux_down_diffuse=fl[1], flux_down_direct=fl[2])
        self._checked(self._lib.rtd_plan_evaluate(self._h, ntau, _lib.dptr(tau), nphi, _lib.dptr(phi),
                                                  int(bool(antiderivative)) | (2 if skip_nt else 0), _lib.dptr(u),
                                                  _lib.dptr(u0),
                                                  _lib.dptr(fl[0]), _lib.dptr(fl[1]), _lib.dptr(fl[2]),
                                                  _lib.dptr(ul)), out)
        return out

    # ---- throughput form (results stay in HBM until fetch) ----
    def set_eval_points(self, tau, phi):
        tau = _f64(np.atleast_2d(tau))
        phi = _f64(np.atleast_1d(phi))
        self._ev_shape = (tau.shape[1], len(phi))
        _lib.check(self._lib.rtd_plan_set_eval_points(self._h, tau.shape[1], _lib.dptr(tau), len(phi),
                                                      _lib.dptr(phi)))

    def run(self):
        _lib.check(self._lib.rtd_plan_run(self._h))
        self.solved = True

    def invalidate_tables(self):
        """Treat the resident inputs as new: the next solve recomputes the per-column Legendre tables at -mu0 and the beam
        attenuations it would otherwise keep from run to run (include/rtd.h: rtd_plan_invalidate_tables)."""
        _lib.check(self._lib.rtd_plan_invalidate_tables(self._h))

    def retained(self):
        """True when evaluate() after a solve only evaluates (one window, or the evaluator state of every column is resident)."""
        r = C.c_int32()
        _lib.check(self._lib.rtd_plan_retained(self._h, C.byref(r)))
        return bool(r.value)

    def retained_form(self):
        """"full" (one window, or everything the evaluators read is resident for every column), "lean" (coefficients, k, E, B
        resident; evaluate() re-runs the eigen stage for the layers its points touch, never the boundary-condition solve) or
        None (evaluate() solves the windows again)."""
        r = C.c_int32()
        _lib.check(self._lib.rtd_plan_retained(self._h, C.byref(r)))
        return {0: None, 1: "full", 2: "lean"}[r.value]

    def windows(self):
        """(columns per window, number of windows) of the plan's work arena."""
        a, b = C.c_int32(), C.c_int32()
        _lib.check(self._lib.rtd_plan_windows(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def run_fetch(self, out=None):
        """run() and fetch() as one host-to-host pipeline (device-to-host copies of a window overlap the next window's
        kernels; include/rtd.h: rtd_plan_run_fetch).  `out`: dict of preallocated arrays to fill (u, u0, flux_up,
        flux_down_diffuse, flux_down_direct; missing keys are not fetched), default: all of them, freshly allocated."""
        ntau, nphi = self._ev_shape
        if out is None:
            out = dict(u=np.empty((self.C, self.Q, ntau, nphi)), u0=np.empty((self.C, self.Q, ntau)),
                       flux_up=np.empty((self.C, ntau)), flux_down_diffuse=np.empty((self.C, ntau)),
                       flux_down_direct=np.empty((self.C, ntau)))
        keys = ("u", "u0", "flux_up", "flux_down_diffuse", "flux_down_direct")
        self._checked(self._lib.rtd_plan_run_fetch(self._h, *[_lib.dptr(out.get(k)) for k in keys]), out)
        self.solved = True
        return out

    def fetch(self):
        ntau, nphi = self._ev_shape
        u = np.empty((self.C, self.Q, ntau, nphi))
        u0 = np.empty((self.C, self.Q, ntau))
        fl = [np.empty((self.C, ntau)) for _ in range(3)]
        out = dict(u=u, u0=u0, flux_up=fl[0], flux_down_diffuse=fl[1], flux_down_direct=fl[2])
        self._checked(self._lib.rtd_plan_fetch(self._h, _lib.dptr(u), _lib.dptr(u0), *[_lib.dptr(a) for a in fl]), out)
        return out

    def result_dev_ptrs(self):
        up, fp = C.c_void_p(), C.c_void_p()
        ub, fb = C.c_int64(), C.c_int64()
        _lib.check(self._lib.rtd_plan_result_dev_ptrs(self._h, C.byref(up), C.byref(ub), C.byref(fp), C.byref(fb)))
        return (up.value, ub.value), (fp.value, fb.value)

    def tensors(self, column=0):
        """The reference's tensors for one column: GC, K, B, G_inv_mu_inv, G."""
        M, L, Q = self.M, self.L, self.Q
        GC, G = np.empty((M, L, Q, Q)), np.empty((M, L, Q, Q))
        K, B, Z = np.empty((M, L, Q)), np.empty((M, L, Q)), np.zeros((L, Q))
        _lib.check(self._lib.rtd_plan_get_tensors(self._h, column, _lib.dptr(GC), _lib.dptr(K), _lib.dptr(B),
                                                  _lib.dptr(Z), _lib.dptr(G)))
        return dict(GC=GC, K=K, B=B, G_inv_mu_inv=Z, G=G)

    # ---- RCCL over xGMI (one rank per plan) ----
    @staticmethod
    def comm_preload():
        _lib.check(_lib.load().rtd_comm_preload())

    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().rtd_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid, rank, nranks):
        _lib.check(self._lib.rtd_comm_init(self._h, uid, rank, nranks))
        self._nranks = nranks

    def comm_size(self):
        """(nranks, rank, device) as RCCL itself reports them for the plan's communicator (ncclCommCount,
        ncclCommUserRank, ncclCommCuDevice; include/rtd.h: rtd_comm_size)."""
        n, r, d = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        _lib.check(self._lib.rtd_comm_size(self._h, C.byref(n), C.byref(r), C.byref(d)))
        return int(n.value), int(r.value), int(d.value)

    def comm_transport(self):
        """Which transport carries the collectives: "rccl", or "stub:ipc" / "stub:shm" when the environment named the tests'
        stand-in (include/rtd.h: rtd_comm_transport)."""
        buf = C.create_string_buffer(32)
        _lib.check(self._lib.rtd_comm_transport(self._h, buf, 32))
        return buf.value.decode()

    def allgather_fluxes(self):
        _lib.check(self._lib.rtd_comm_allgather_fluxes(self._h))

    def solve_layers(self, first, count):
        """Eigen stage of the layers [first, first + count) only (layer shards; include/rtd.h: rtd_plan_solve_layers)."""
        _lib.check(self._lib.rtd_plan_solve_layers(self._h, int(first), int(count)))
        self.solved = False

    def allgather_layers(self, count_per_rank):
        """ONE RCCL all-gather of the eigen-stage results of every rank's layers (rtd_comm_allgather_layers)."""
        _lib.check(self._lib.rtd_comm_allgather_layers(self._h, int(count_per_rank)))

    def solve_bc(self):
        """Boundary-condition solve over all layers, after solve_layers / allgather_layers."""
        _lib.check(self._lib.rtd_plan_solve_bc(self._h))
        self.solved = True

    def allgather_results(self):
        """RCCL all-gather of u and the fluxes of the last run() over the ranks, on the plan's communication stream
        (overlaps the next run(); include/rtd.h: rtd_comm_allgather_results)."""
        _lib.check(self._lib.rtd_comm_allgather_results(self._h))

    def gather_results(self, root=0):
        """The results of the last run() gathered on rank `root` only (ncclSend / ncclRecv; include/rtd.h:
        rtd_comm_gather_results); every rank calls it."""
        _lib.check(self._lib.rtd_comm_gather_results(self._h, int(root)))

    def fetch_gathered_results(self, want_u=True):
        """-> (u [nranks * C, Q, ntau, nphi] or None, fluxes [nranks, 3, C, ntau]) of the last allgather_results()."""
        ntau, nphi = self._ev_shape
        u = np.empty((self._nranks * self.C, self.Q, ntau, nphi)) if want_u else None
        fl = np.empty((self._nranks, 3, self.C, ntau))
        _lib.check(self._lib.rtd_comm_fetch_gathered_results(self._h, _lib.dptr(u), _lib.dptr(fl)))
        return u, fl

    def fetch_gathered_columns(self, rank, first, count, want_u=True):
        """-> (u [count, Q, ntau, nphi] or None, fluxes [3, count, ntau]): the columns [first, first + count) of rank
        `rank`'s shard in the gathered arrays of the last allgather_results() / gather_results()."""
        ntau, nphi = self._ev_shape
        u = np.empty((count, self.Q, ntau, nphi)) if want_u else None
        fl = np.empty((3, count, ntau))
        _lib.check(self._lib.rtd_comm_fetch_gathered_columns(self._h, int(rank), int(first), int(count), _lib.dptr(u), _lib.dptr(fl)))
        return u, fl

    def fetch_gathered(self):
        out = np.empty((self._nranks, 3, self.C, self._ev_shape[0]))
        _lib.check(self._lib.rtd_comm_fetch_gathered(self._h, _lib.dptr(out)))
        return out

    def enable_timing(self, on=True):
        _lib.check(self._lib.rtd_plan_enable_timing(self._h, int(on)))

    def timing(self, reset=True):
        ms = (C.c_double * 7)()
        n = (C.c_int64 * 7)()
        _lib.check(self._lib.rtd_plan_get_timing(self._h, ms, n, int(reset)))
        names = ("tables", "asm", "jacobi", "post", "iface", "sweep", "eval")
        return {k: (ms[i], n[i]) for i, k in enumerate(names)}

    def pivoted_chains(self):
        """(column, mode) chains of the last window that the tiled 64-stream BC kernel handed to the pivoted kernels."""
        n = C.c_int32()
        _lib.check(self._lib.rtd_plan_pivoted_chains(self._h, C.byref(n)))
        return n.value

    # ---- a failed column in a batch: raise (default) or mark ----
    numeric_errors = "raise"  # or "nan": fill the failed columns of the returned arrays with NaN instead of raising

    def _checked(self, rc, arrays):
        """Maps the status of a call that returns results.  numeric_errors == "nan": a numerical failure (rc 5) is not raised;
        the columns it belongs to are NaN in `arrays` (u: any failed mode; u0 and fluxes: a failed mode 0 -- the others
        keep their values: they come from mode 0 alone), every other column keeps its result."""
        if rc == _lib.RTD_ERR_NUMERIC and self.numeric_errors == "nan":
            st = self.column_status()
            for k, a in arrays.items():
                if a is not None:
                    a[(st & (0xFFFF if k in ("u", "ulast") else 0xFF)) != 0] = np.nan
            return
        _lib.check(rc)

    def column_status(self):
        """int32 [C]: 0 for a column that solved cleanly, else the numerical-failure bits of include/rtd.h
        (rtd_plan_get_column_status): low byte raised by Fourier mode 0, next byte by modes m > 0 (fluxes and u0 of such a
        column are still valid).  After a NumericalError from a batch this tells which columns to drop."""
        out = np.zeros(self.C, dtype=np.int32)
        _lib.check(self._lib.rtd_plan_get_column_status(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def max_sweeps(self):
        s = C.c_int32()
        _lib.check(self._lib.rtd_plan_max_sweeps(self._h, C.byref(s)))
        return s.value


def device_count():
    n = C.c_int32()
    lib = _lib.load()
    rc = lib.rtd_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _inputs_struct(prep):
    """(rtd_inputs, keep-alive list) for the one-call entry points of include/rtd.h."""
    keys = [("mu_pos", "mu"), ("weights", "W"), ("scaled_omega", "omega_s"), ("tau", "tau"),
            ("scaled_tau_with_0", "tau_s0"), ("scale_tau", "scale_tau"), ("wleg", "wleg"), ("mu0", "mu0"), ("I0", "I0"),
            ("phi0", "phi0"), ("rescale", "rescale"), ("b_pos", "b_pos"), ("b_neg", "b_neg"), ("s_poly", "s_s"),
            ("bdrf_q", "bdrf_q"), ("bdrf_q0", "bdrf_q0")]
    keep, s = [], _lib.rtd_inputs()
    for field, k in keys:
        a = _f64(prep[k])
        keep.append(a)
        setattr(s, field, _lib.dptr(a))
    return s, keep


def _dims(prep):
    return _lib.rtd_dims(prep["C"], prep["L"], 2 * prep["N"], prep["P"], prep["M"], prep["Ns"], prep["NBDRF"],
                         1 if prep["beam"] else 0)


def solve_batch_once(prep, tau, phi, device=0):
    """One call of the C ABI's ``rtd_solve_batch``: plan creation, upload, solve, evaluation and teardown inside the
    library.  tau [C, ntau], phi [nphi] -> dict(u, u0, flux_up, flux_down_diffuse, flux_down_direct)."""
    lib = _lib.load()
    s, keep = _inputs_struct(prep)
    dims = _dims(prep)
    tau, phi = _f64(np.atleast_2d(tau)), _f64(np.atleast_1d(phi))
    Cn, Q, nt, nph = prep["C"], 2 * prep["N"], tau.shape[1], phi.shape[0]
    out = dict(u=np.empty((Cn, Q, nt, nph)), u0=np.empty((Cn, Q, nt)), flux_up=np.empty((Cn, nt)),
               flux_down_diffuse=np.empty((Cn, nt)), flux_down_direct=np.empty((Cn, nt)))
    _lib.check(lib.rtd_solve_batch(C.byref(dims), device, C.byref(s), nt, _lib.dptr(tau), nph, _lib.dptr(phi),
                                   *[_lib.dptr(out[k]) for k in ("u", "u0", "flux_up", "flux_down_diffuse", "flux_down_direct")]))
    return out


def solve_tensors_once(prep, column=0, device=0):
    """One call of ``rtd_solve_tensors``: the tensors the reference's closures capture for one column."""
    lib = _lib.load()
    s, keep = _inputs_struct(prep)
    dims = _dims(prep)
    M, L, Q = prep["M"], prep["L"], 2 * prep["N"]
    out = dict(GC=np.empty((M, L, Q, Q)), K=np.empty((M, L, Q)), B=np.empty((M, L, Q)), G_inv_mu_inv=np.empty((L, Q)),
               G=np.empty((M, L, Q, Q)))
    _lib.check(lib.rtd_solve_tensors(C.byref(dims), device, C.byref(s), column,
                                     *[_lib.dptr(out[k]) for k in ("GC", "K", "B", "G_inv_mu_inv", "G")]))
    return out
